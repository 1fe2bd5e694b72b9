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


# ---- condensed form of the batched handle (round 5) -----------------------------------------------------
def _cond_shapes():
    return [(3, 0, 9), (64, 0, 64), (65, 63, 1), (130, 40, 100), (200, 0, 400), (256, 0, 256), (500, 20, 250), (100, 30, 447)]


@pytest.mark.parametrize("n,me,mi", _cond_shapes())
def test_batched_condensed_vs_oracle_and_full_form(n, me, mi):
    """set_option("condensed", 1) on a batched handle: n + me + |A| columns factored per problem, the full direction and the
    inertia of the FULL matrix returned; against the oracle's LU, against the full form, and the device-side backward error
    against a torch evaluation."""
    import torch
    from pyipm_amd.batched import BatchedNewton
    B = 5
    qps = [make_qp(n, me, mi, seed=900 + 7 * b + n) for b in range(B)]
    opt = lambda key, on: _stack(qps, key) if on else None          # noqa: E731
    args = (_stack(qps, "d2L"), opt("Je", me), opt("Ji", mi), _stack(qps, "df"), opt("ce", me), opt("ci", mi), opt("s", mi),
            opt("lam", me + mi))
    full = BatchedNewton(n, me, mi)
    dz0, st0 = full.step_all(*args, mu=0.2)
    cond = BatchedNewton(n, me, mi, condensed=True)
    dz1, st1 = cond.step_all(*args, mu=0.2)
    assert cond.n_condensed_fallback == 0
    be = cond.last_backward_errors.cpu().numpy()
    assert be.max() <= 1e-12
    dz0, dz1 = dz0.cpu().numpy(), dz1.cpu().numpy()
    N = n + 2 * mi + me
    for b, q in enumerate(qps):
        ref, _, Hc, g = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                        q["mu"], n, me, mi, regularise=False)
        tol = max(1e-10, 20 * np.linalg.cond(Hc) * np.finfo(float).eps)
        assert np.linalg.norm(dz1[b] - ref) / np.linalg.norm(ref) <= tol
        assert np.linalg.norm(dz1[b] - dz0[b]) / np.linalg.norm(dz0[b]) <= tol
        assert (st1[b]["n_neg"], st1[b]["n_pos"], st1[b]["n_zero"]) == (me + mi, N - me - mi, 0)
        raw = dz1[b].copy(); raw[n + mi:] *= -1.0
        assert abs(np.linalg.norm(Hc @ raw - g) / np.linalg.norm(g) - be[b]) <= 1e-13      # the device's own backward error
    # the full form's directions through the same device check (1e-10: me ~ n makes some of these systems ill-conditioned)
    assert float(full.backward_errors(torch.from_numpy(dz0).cuda()).max()) <= 1e-10
    full.close(); cond.close()


def test_config5_condensed_512_problems():
    """BASELINE config 5 through the condensed form: 512 x (256, 0, 256) -> 256 columns factored per problem instead of 768;
    every 32nd problem against the oracle, all of them through the backward error against the full blocks and the inertia
    of the full matrix."""
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
    ci = s + 0.1 * torch.randn(B, mi, dtype=f64, device=dev, generator=gen)
    Ji = G.transpose(1, 2).contiguous()
    bn = BatchedNewton(n, me, mi, condensed=True)
    dz, stats = bn.step_all(Q, None, Ji, c, None, ci, s, lam, mu=0.2)
    assert bn.n_condensed_fallback == 0
    assert all(st["n_neg"] == mi and st["n_pos"] == n + mi and st["n_zero"] == 0 for st in stats)
    assert float(bn.last_backward_errors.max()) <= 1e-12
    for b in range(0, B, 32):
        ref, _, _, _ = orc.newton_step(Q[b].cpu().numpy(), None, Ji[b].cpu().numpy(), c[b].cpu().numpy(), None,
                                       ci[b].cpu().numpy(), s[b].cpu().numpy(), lam[b].cpu().numpy(), 0.2, n, me, mi,
                                       regularise=False)
        assert np.linalg.norm(dz[b].cpu().numpy() - ref) / np.linalg.norm(ref) <= 1e-10
    tm = bn.last_ms()
    assert tm["step_ms"] > 0.0 and tm["factor_ms"] > 0.0
    bn.close()


@pytest.mark.parametrize("decades", [6, 12, 20])
def test_batched_condensed_sigma_spread(decades):
    """Late in an interior-point run Sigma = lda / s spans many decades: the pairs above condensed_sigma_max stay explicit rows
    (another count in every problem of the batch), the direction still matches the oracle's LU of the FULL system."""
    from pyipm_amd.batched import BatchedNewton
    n, me, mi, B = 120, 20, 90, 6
    rng = np.random.default_rng(decades)
    qps = [make_qp(n, me, mi, seed=1200 + b) for b in range(B)]
    for b, q in enumerate(qps):
        k = rng.permutation(mi)[: 10 + 9 * b]                     # a different active set per problem
        e = rng.uniform(-decades / 2.0, decades / 2.0, size=k.size)
        q["s"][k] = 10.0 ** (-e / 2.0); q["lam"][me + k] = 10.0 ** (e / 2.0)
    args = tuple(_stack(qps, key) for key in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam"))
    bn = BatchedNewton(n, me, mi, condensed=True)
    dz, st = bn.step_all(*args, mu=0.2)
    dz = dz.cpu().numpy()
    assert bn.n_condensed_fallback == 0
    for b, q in enumerate(qps):
        ref, _, Hc, _ = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                        q["mu"], n, me, mi, regularise=False)
        tol = max(1e-10, 50 * np.linalg.cond(Hc) * np.finfo(float).eps)
        assert np.linalg.norm(dz[b] - ref) / np.linalg.norm(ref) <= tol
        assert st[b]["n_neg"] == me + mi and st[b]["n_zero"] == 0
    bn.close()


def test_batched_condensed_guard_falls_back():
    """Non-convex members (indefinite d2L) in a condensed batch: whatever the condensed factor makes of them, what comes back
    has the eigen-inertia of the full matrix and the oracle's direction -- through the condensed form when it passes the
    device-side check, through the full form when it does not (n_condensed_fallback says which)."""
    from pyipm_amd.batched import BatchedNewton
    n, me, mi, B = 70, 10, 30, 8
    rng = np.random.default_rng(11)
    qps = [make_qp(n, me, mi, seed=1300 + b) for b in range(B)]
    for b in range(0, B, 2):
        M = rng.standard_normal((n, n))
        qps[b]["d2L"] = (M + M.T) / 2
    bn = BatchedNewton(n, me, mi, condensed=True)
    dz, stats = bn.step_all(*[_stack(qps, k) for k in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam")], mu=0.2)
    dz = dz.cpu().numpy()
    for b, q in enumerate(qps):
        H = orc.kkt_matrix(q["d2L"], q["Je"], q["Ji"], q["s"], q["lam"], n, me, mi)
        assert stats[b]["n_neg"] == int((np.linalg.eigvalsh(H) < 0).sum())
        ref, _, _, _ = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"], q["lam"],
                                       q["mu"], n, me, mi, regularise=False)
        assert np.linalg.norm(dz[b] - ref) / np.linalg.norm(ref) <= 1e-8
    # a bar nothing can meet: the guard must take the full form and still return the right directions
    bn.condensed_tol = 0.0
    before = bn.n_condensed_fallback
    dz2, _ = bn.step_all(*[_stack(qps, k) for k in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam")], mu=0.2)
    assert bn.n_condensed_fallback == before + 1
    full = BatchedNewton(n, me, mi)
    dz3, _ = full.step_all(*[_stack(qps, k) for k in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam")], mu=0.2)
    assert np.array_equal(dz2.cpu().numpy(), dz3.cpu().numpy())
    bn.close(); full.close()


@pytest.mark.parametrize("n,me,mi,spread", [(64, 0, 32, 0), (128, 10, 96, 0), (192, 0, 160, 0), (256, 0, 256, 0), (256, 20, 64, 0),
                                            (128, 0, 128, 12), (256, 30, 224, 12)])
def test_batched_condensed_gram_per_problem_gives_the_same_bits(n, me, mi, spread):
    """k_bc_assemble_p (one workgroup per problem: the Jacobian chunk of all n columns staged once for every x-x tile) against
    k_bc_assemble (a workgroup per tile) where the former applies -- n a multiple of 64 up to 256, mi a multiple of 32 --: the
    condensed matrices, hence factors, statistics and directions, are bit for bit the same; with `spread` decades of Sigma some
    inequalities stay explicit rows (the tile rows below the x block).  Block structure: pyipm.py:824-842."""
    import torch
    from pyipm_amd.batched import BatchedNewton
    B = 12
    qps = [make_qp(n, me, mi, seed=900 + b) for b in range(B)]
    if spread:
        rng = np.random.default_rng(n + mi)
        for q in qps:
            q["lam"] = q["lam"].copy()
            q["lam"][me:] = q["s"] * 10.0 ** rng.uniform(-spread / 2, spread / 2, mi)
    args = [_stack(qps, k) for k in ("d2L", "Je", "Ji", "df", "ce", "ci", "s", "lam")]
    out = {}
    for per_problem in (1, 0):
        bn = BatchedNewton(n, me, mi, condensed=True, guard=False)
        bn.set_option("expert", 1)
        bn.set_option("bc_per_problem", per_problem)
        dz, stats = bn.step_all(*args, mu=0.2)
        out[per_problem] = (dz.clone(), [{k: v for k, v in st.items()} for st in stats])
        bn.close()
    assert torch.equal(out[1][0], out[0][0])
    assert out[1][1] == out[0][1]
    for b in (0, B - 1):
        q = qps[b]
        ref, _, _, _ = orc.newton_step(q["d2L"], q["Je"], q["Ji"], q["df"], q["ce"], q["ci"], q["s"],
                                       args[7][b] if spread else q["lam"], 0.2, n, me, mi, regularise=False)
        assert np.linalg.norm(out[1][0][b].cpu().numpy() - ref) / np.linalg.norm(ref) <= 1e-9
