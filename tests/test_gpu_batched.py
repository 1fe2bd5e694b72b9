"""BASELINE.json configs[4]: batched independent QPs (multi-start) — replicas, no exchange."""
import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import make_qp

pytestmark = pytest.mark.gpu


def _stack(qps, key):
    return np.stack([q[key] for q in qps])


def test_batched_small_vs_oracle():
    from pyipm_amd.batched import BatchedNewton
    n, me, mi, B = 48, 8, 24, 20
    qps = [make_qp(n, me, mi, seed=100 + b) for b in range(B)]
    bn = BatchedNewton(n, me, mi, workers=4)
    dz, stats = bn.step_all(_stack(qps, "d2L"), _stack(qps, "Je"), _stack(qps, "Ji"), _stack(qps, "df"),
                            _stack(qps, "ce"), _stack(qps, "ci"), _stack(qps, "s"), _stack(qps, "lam"), mu=0.2)
    dz = dz.cpu().numpy()
    for b, q in enumerate(qps):
        ref, _, _, _ = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                       q["mu"], n, me, mi, regularise=False)
        assert np.linalg.norm(dz[b] - ref) / np.linalg.norm(ref) <= 1e-10
        assert stats[b]["n_neg"] == me + mi and stats[b]["n_zero"] == 0
    bn.close()


def _batched_shapes():
    """Block boundaries on and off the 64-tile grid, empty blocks, one tile, the 1024-row limit of a batched problem;
    PYIPM_BATCHED_FUZZ=count[,seed] adds seeded random shapes by hand."""
    import os
    out = [(1, 0, 0), (5, 2, 0), (3, 0, 9), (64, 0, 64), (65, 63, 1), (130, 40, 100), (200, 0, 400), (500, 20, 250), (1020, 0, 2)]
    spec = os.environ.get("PYIPM_BATCHED_FUZZ", "")
    if spec:
        count, seed = (spec.split(",") + ["3"])[:2]
        rng = np.random.default_rng(int(seed))
        while len(out) < 9 + int(count):
            n = int(rng.integers(1, 500))
            me = 0 if rng.random() < 0.3 else int(rng.integers(1, n + 1))
            mi = 0 if rng.random() < 0.3 else int(rng.integers(1, 300))
            if n + me + 2 * mi <= 1024:
                out.append((n, me, mi))
    return out


@pytest.mark.parametrize("n,me,mi", _batched_shapes())
def test_batched_ragged_shapes_vs_oracle(n, me, mi):
    from pyipm_amd.batched import BatchedNewton
    B = 5
    qps = [make_qp(n, me, mi, seed=700 + 13 * b + n) for b in range(B)]
    bn = BatchedNewton(n, me, mi, workers=2)
    opt = lambda key, on: _stack(qps, key) if on else None          # noqa: E731
    dz, stats = bn.step_all(_stack(qps, "d2L"), opt("Je", me), opt("Ji", mi), _stack(qps, "df"), opt("ce", me), opt("ci", mi),
                            opt("s", mi), opt("lam", me + mi), mu=0.2)
    dz = dz.cpu().numpy()
    for b, q in enumerate(qps):
        ref, _, Hc, _ = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                        q["mu"], n, me, mi, regularise=False)
        tol = max(1e-10, 20 * np.linalg.cond(Hc) * np.finfo(float).eps)
        assert np.linalg.norm(dz[b] - ref) / np.linalg.norm(ref) <= tol
        assert stats[b]["n_neg"] == me + mi and stats[b]["n_zero"] == 0
    bn.close()


def test_config5_512_problems():
    """512 independent n=256 QPs with 256 inequalities each (KKT dim 768); every 32nd one against the oracle,
    all of them through inertia + linear-system residual computed with torch fp64."""
    import torch
    from pyipm_amd.batched import BatchedNewton
    n, me, mi, B = 256, 0, 256, 512
    gen = torch.Generator(device="cuda").manual_seed(5)
    f64, dev = torch.float64, "cuda"
    M = torch.randn(B, n, n, dtype=f64, device=dev, generator=gen)
    Q = M @ M.transpose(1, 2) / n + torch.eye(n, dtype=f64, device=dev)
    G = torch.randn(B, mi, n, dtype=f64, device=dev, generator=gen) / np.sqrt(n)
    c = torch.randn(B, n, dtype=f64, device=dev, generator=gen)
    s = torch.rand(B, mi, dtype=f64, device=dev, generator=gen) * 1.5 + 0.5
    lam = torch.rand(B, mi, dtype=f64, device=dev, generator=gen) * 1.5 + 0.5
    ci = s + 0.1 * torch.randn(B, mi, dtype=f64, device=dev, generator=gen)        # small ci - s residual
    Ji = G.transpose(1, 2).contiguous()
    bn = BatchedNewton(n, me, mi, workers=16)
    dz, stats = bn.step_all(Q, None, Ji, c, None, ci, s, lam, mu=0.2)
    assert all(st["n_neg"] == mi and st["n_zero"] == 0 for st in stats)
    eps = np.finfo(float).eps
    # verify H dz' = g block-wise in torch (dz has the multiplier block sign-flipped: undo it)
    dx, ds, dl = dz[:, :n], dz[:, n:n + mi], -dz[:, n + mi:]
    gx = -(c - torch.einsum("bnm,bm->bn", Ji, lam))
    gs = -(lam - 0.2 / (s + eps))
    gl = -(ci - s)
    r1 = torch.einsum("bij,bj->bi", Q, dx) + torch.einsum("bnm,bm->bn", Ji, dl) - gx
    r2 = lam / (s + eps) * ds - dl - gs
    r3 = torch.einsum("bnm,bn->bm", Ji, dx) - ds - gl
    res = torch.sqrt((r1 ** 2).sum(1) + (r2 ** 2).sum(1) + (r3 ** 2).sum(1))
    nrm = torch.sqrt((gx ** 2).sum(1) + (gs ** 2).sum(1) + (gl ** 2).sum(1))
    assert float((res / nrm).max()) <= 1e-12
    for b in range(0, B, 32):
        ref, _, _, _ = orc.newton_step(Q[b].cpu().numpy(), None, Ji[b].cpu().numpy(), c[b].cpu().numpy(), None,
                                       ci[b].cpu().numpy(), s[b].cpu().numpy(), lam[b].cpu().numpy(), 0.2, n, me, mi,
                                       regularise=False)
        assert np.linalg.norm(dz[b].cpu().numpy() - ref) / np.linalg.norm(ref) <= 1e-10
    bn.close()


def test_batched_indefinite_inertia_and_guards():
    """Non-convex members of a batch (indefinite d2L -> 2x2 / off-diagonal pivots, flagged tiles) report the
    eigen-inertia of THEIR matrix; a batch of one works; single-system entry points refuse a batched handle."""
    import torch
    from pyipm_amd.batched import BatchedNewton
    from pyipm_amd.newton import NewtonError, FactorStats
    import ctypes
    n, me, mi, B = 70, 10, 30, 9
    rng = np.random.default_rng(3)
    qps = [make_qp(n, me, mi, seed=300 + b) for b in range(B)]
    for b in range(0, B, 2):                                     # every other problem gets an indefinite Hessian
        M = rng.standard_normal((n, n))
        qps[b]["d2L"] = (M + M.T) / 2
    bn = BatchedNewton(n, me, mi)
    dz, stats = bn.step_all(_stack(qps, "d2L"), _stack(qps, "Je"), _stack(qps, "Ji"), _stack(qps, "df"),
                            _stack(qps, "ce"), _stack(qps, "ci"), _stack(qps, "s"), _stack(qps, "lam"), mu=0.2)
    dz = dz.cpu().numpy()
    for b, q in enumerate(qps):
        H = orc.kkt_matrix(q["d2L"], q["Je"], q["Ji"], q["s"], q["lam"], n, me, mi)
        w = np.linalg.eigvalsh(H)
        assert stats[b]["n_neg"] == int((w < 0).sum()) and stats[b]["n_zero"] == 0
        ref, _, _, g = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                       q["mu"], n, me, mi, regularise=False)
        assert np.linalg.norm(dz[b] - ref) / np.linalg.norm(ref) <= 1e-9
    assert any(st["n_2x2"] > 0 for st in stats[0::2])
    one, st1 = bn.step_all(*[_stack(qps[:1], k) for k in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam")], mu=0.2)
    assert np.array_equal(one.cpu().numpy()[0], dz[0]) and st1[0]["n_neg"] == stats[0]["n_neg"]
    rc = bn.lib.pyipm_newton_assemble(bn.h, 0.0, 0.0)           # single-system call on a batched handle
    assert rc == -1 and b"batched handle" in bn.lib.pyipm_newton_last_error(bn.h)
    with pytest.raises(NewtonError):
        BatchedNewton(600, 0, 300, batch=2)                      # N = 1200 > 1024: not a small system
    bn.close()
