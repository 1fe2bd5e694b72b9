"""L-BFGS search direction through the C-ABI (include/pyipm_lbfgs.h) against the unmodified reference's
recorded directions (tests/golden/lbfgs_*.npz) and, at sizes the reference never ran, against the CPU oracle
(oracle/lbfgs_oracle.py, itself pinned to those records by tests/test_oracle_golden.py)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = float(np.finfo(float).eps)


def _core(n, me, mi, cap, **kw):
    from pyipm_amd.lbfgs import LbfgsCore
    return LbfgsCore(n, me, mi, cap, device=0, **kw)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "lbfgs_dir_*.npz"))), ids=os.path.basename)
def test_direction_matches_reference_fixture(path):
    d = np.load(path)
    n, me, mi, m = int(d["n"]), int(d["me"]), int(d["mi"]), int(d["m"])
    reg = float(d["reg_coef"]) * float(d["eta"]) * float(d["mu"]) ** float(d["beta"])
    core = _core(n, me, mi, max(m, 1))
    core.stage_jacobian(d["Je"] if me else None, d["Ji"] if mi else None)
    dz, st = core.direction(d["g"], d["s"], d["lda"], float(d["zeta"]), d["S"], d["Y"], d["SS"], d["L"], d["D"], reg=reg)
    dz = dz.cpu().numpy()
    if int(d["rank_deficient"]):
        # duplicated equality gradient: G_ee is singular, the reference adds reg ~ 8e-13 and gets |dz| ~ 4e11;
        # the factorisation must notice (rejected / negative pivot) and take the same branch.  The direction is
        # then the solution of a system of condition ~1e13: compare through the system itself.
        assert st["regularised"] == 1 and st["n_factor"] == 2
        assert np.all(np.isfinite(dz))
        # The value, component by component.  Only the DIFFERENCE of the two duplicated multipliers is ill-determined:
        # (r_0 - r_last) / (2 reg) ~ 3e11, a quotient by the pivot reg + O(eps |G|), which the reference's LU and this
        # factorisation round differently (measured 1e-4 apart).  Their sum and every other entry of the direction are
        # well-determined and carry only the contamination eps x 3e11 ~ 3e-5 of that component.
        ref = d["dz_raw"]
        i0, i1 = n + mi, n + mi + me - 1
        assert abs(ref[i0]) > 1e11 and abs(ref[i0] + ref[i1]) < 10.0                  # (the fixture is what this test thinks it is)
        assert abs(dz[i0] - ref[i0]) <= 1e-3 * abs(ref[i0]) and abs(dz[i1] - ref[i1]) <= 1e-3 * abs(ref[i1])
        assert abs((dz[i0] + dz[i1]) - (ref[i0] + ref[i1])) <= 1e-3
        rest = np.ones(len(ref), dtype=bool)
        rest[[i0, i1]] = False
        assert _rel(dz[rest], ref[rest]) <= 1e-4
        assert _rel(dz, ref) <= 1e-3
    else:
        assert st["regularised"] == 0 and st["n_neg"] == 0 and st["n_zero"] == 0
        assert _rel(dz, d["dz_raw"]) <= 1e-9
    flipped, _ = core.direction(d["g"], d["s"], d["lda"], float(d["zeta"]), d["S"], d["Y"], d["SS"], d["L"], d["D"],
                                reg=reg, flip=True)
    flipped = flipped.cpu().numpy()
    np.testing.assert_array_equal(flipped[:n + mi], dz[:n + mi])
    np.testing.assert_array_equal(flipped[n + mi:], -dz[n + mi:])            # pyipm.py:1723-1725
    core.close()


@pytest.mark.parametrize("k", range(1, 11))
def test_direction_matches_reference_traces(k):
    """Every direction of the reference's lbfgs=4 runs on its example problems, from the recorded inputs."""
    from pyipm_amd.problems import example_problem
    d = np.load(os.path.join(GOLD, "lbfgs_trace_p%02d.npz" % k))
    prob = example_problem(k)
    n, me, mi = int(d["nvar"]), int(d["neq"]), int(d["nineq"])
    core = _core(n, me, mi, int(d["memory"]) + 1)
    for it in range(int(d["n_iter"])):
        m = int(d["it_m"][it])
        x = d["it_x"][it]
        Je = np.asarray(prob["dce"](x), dtype=float).reshape(n, me) if me else None
        Ji = np.asarray(prob["dci"](x), dtype=float).reshape(n, mi) if mi else None
        core.stage_jacobian(Je, Ji)
        reg = np.sqrt(EPS) * 1.0e-4 * float(d["it_mu"][it]) ** 0.4
        dz, st = core.direction(d["it_g"][it], d["it_s"][it], d["it_lda"][it], float(d["it_zeta"][it]),
                                d["it_S"][it][:, :m], d["it_Y"][it][:, :m], d["it_SS"][it][:m, :m],
                                d["it_L"][it][:m, :m], d["it_D"][it][:m, :m], reg=reg)
        ref = d["it_dz_raw"][it]
        assert st["m"] == m
        assert _rel(dz.cpu().numpy(), ref) <= 1e-7, (k, it, st)
    core.close()


@pytest.mark.parametrize("k", range(1, 11))
def test_ipm_lbfgs_solves_example_problems_on_the_device(k):
    """IPM(lbfgs=4) with the HIP backend: the reference's own acceptance test (unit_tests.py:49-51, 405-415:
    |x - x_gt| <= 1e-3) and its termination signal."""
    from pyipm_amd.ipm import IPM
    from pyipm_amd.problems import example_problem, unit_test_x0
    d = np.load(os.path.join(GOLD, "lbfgs_trace_p%02d.npz" % k))
    p = example_problem(k)
    ipm = IPM(x0=unit_test_x0()[k], f=p["f"], df=p["df"], ce=p["ce"], dce=p["dce"], ci=p["ci"], dci=p["dci"],
              lbfgs=4, Ftol=1.0e-8, verbosity=-1, device=0)
    with np.errstate(all="ignore"):
        x, s, lda, fval, kkt = ipm.solve()
    assert min(np.linalg.norm(x - gt) for gt in p["ground_truth"]) <= 1e-3
    assert ipm.signal == int(d["signal"])
    assert ipm.backend.n_calls == ipm.iter_count
    np.testing.assert_allclose(x, d["x"], rtol=1e-5, atol=1e-6)


def _storage(n, m, rng, constrained, memory):
    """Displacement storage built with the oracle's restatement of lbfgs_update (pyipm.py:1282-1371)."""
    from oracle import lbfgs_oracle as lo
    zeta, S, Y, SS, L, D, fail = lo.lbfgs_init(n)
    Mq = rng.standard_normal((n, 32)) / np.sqrt(32.0)
    x_old = rng.standard_normal(n)
    for _ in range(m):
        x_new = x_old + rng.standard_normal(n) / np.sqrt(n)
        hv = lambda v: Mq @ (Mq.T @ v) + 0.5 * v           # noqa: E731   SPD curvature model, applied matrix-free
        zeta, S, Y, SS, L, D, fail = lo.lbfgs_update(x_old, x_new, -hv(x_old), -hv(x_new), zeta, S, Y, SS, L, D, fail,
                                                     n, constrained, memory, EPS)
        x_old = x_new
    return zeta, S, Y, SS, L, D


@pytest.mark.parametrize("n,me,mi,m", [(1500, 0, 0, 7), (1000, 130, 0, 5), (1200, 0, 333, 6), (2000, 100, 300, 8),
                                       (777, 65, 191, 31), (3000, 256, 512, 2), (5, 1, 2, 3), (300, 64, 64, 0)])
def test_direction_matches_oracle_at_larger_sizes(n, me, mi, m):
    from oracle import lbfgs_oracle as lo
    from pyipm_amd.problems import make_qp
    rng = np.random.default_rng(n + 7 * me + 13 * mi + m)
    qp = make_qp(n, me, mi, 3)
    zeta, S, Y, SS, L, D = _storage(n, m, rng, bool(me or mi), max(m, 1))
    assert S.shape[1] == m
    g = rng.standard_normal(n + 2 * mi + me)
    s = qp["s"] if mi else np.zeros(0)
    lda = qp["lam"] if (me or mi) else np.zeros(0)
    ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=qp["Je"] if me else None, Ji=qp["Ji"] if mi else None, s=s, lda=lda,
                       reg=1e-12)
    core = _core(n, me, mi, max(m, 1), nb=128 if n < 1000 else 256)
    core.stage_jacobian(qp["Je"] if me else None, qp["Ji"] if mi else None)
    dz, st = core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=1e-12)
    assert st["regularised"] == 0 and st["n_neg"] == 0 and st["n_zero"] == 0
    assert _rel(dz.cpu().numpy(), ref) <= 1e-9
    tm = core.last_timings()
    assert tm["total_ms"] > 0.0
    core.close()


def test_sigma_spread_and_device_inputs():
    """Sigma over 16 orders of magnitude (late interior-point iterations) and device-tensor inputs."""
    import torch
    from oracle import lbfgs_oracle as lo
    from pyipm_amd.problems import make_qp
    n, me, mi, m = 600, 40, 200, 4
    rng = np.random.default_rng(99)
    qp = make_qp(n, me, mi, 8)
    zeta, S, Y, SS, L, D = _storage(n, m, rng, True, m)
    sig = np.exp(rng.uniform(np.log(1e-8), np.log(1e8), mi))
    s = rng.uniform(0.5, 2.0, mi)
    lda = np.concatenate([qp["lam"][:me], sig * s])
    g = rng.standard_normal(n + 2 * mi + me)
    ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=qp["Je"], Ji=qp["Ji"], s=s, lda=lda, reg=1e-12)
    core = _core(n, me, mi, m)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()       # noqa: E731
    core.stage_jacobian(dev(qp["Je"]), dev(qp["Ji"]))
    dz, st = core.direction(dev(g), dev(s), dev(lda), zeta, dev(S), dev(Y), SS, L, D, reg=1e-12)
    assert st["regularised"] == 0
    # the s rows are scaled by 1/Sigma (up to 1e8): compare block-wise relative to each block's own size
    out = dz.cpu().numpy()
    for lo_, hi in ((0, n), (n, n + mi), (n + mi, n + mi + me), (n + mi + me, n + 2 * mi + me)):
        assert _rel(out[lo_:hi], ref[lo_:hi]) <= 1e-7
    core.close()


def test_argument_errors_are_reported_not_thrown():
    from pyipm_amd.lbfgs import LbfgsCore
    from pyipm_amd.newton import NewtonError
    with pytest.raises(NewtonError):
        LbfgsCore(10, 0, 0, 33, device=0)                      # max_pairs <= 32
    core = _core(10, 2, 0, 2)
    z = np.zeros
    with pytest.raises(NewtonError, match="stage the Jacobians"):
        core.direction(z(12), z(0), z(2), 1.0, z((10, 0)), z((10, 0)), z((0, 0)), z((0, 0)), z((0, 0)))
    core.stage_jacobian(np.eye(10)[:, :2], None)
    with pytest.raises(NewtonError, match="max_pairs"):
        core.direction(z(12), z(0), z(2), 1.0, z((10, 3)), z((10, 3)), z((3, 3)), z((3, 3)), z((3, 3)))
    with pytest.raises(NewtonError, match="zeta"):
        core.direction(z(12), z(0), z(2), 0.0, z((10, 0)), z((10, 0)), z((0, 0)), z((0, 0)), z((0, 0)))
    core.close()


def test_gram_matrix_is_reused_until_the_jacobians_are_staged_again():
    """J'J depends on neither zeta nor Sigma: directions with different zeta / s / lda between two stagings must
    reuse it (one Gram launch) and still match the oracle; a new staging must invalidate it."""
    from oracle import lbfgs_oracle as lo
    from pyipm_amd.problems import make_qp
    n, me, mi, m = 900, 50, 150, 4
    rng = np.random.default_rng(5)
    qp = make_qp(n, me, mi, 4)
    zeta, S, Y, SS, L, D = _storage(n, m, rng, True, m)
    core = _core(n, me, mi, m)
    core.stage_jacobian(qp["Je"], qp["Ji"])
    for trial in range(3):
        g = rng.standard_normal(n + 2 * mi + me)
        s = rng.uniform(0.5, 2.0, mi)
        lda = np.concatenate([rng.standard_normal(me), rng.uniform(0.1, 3.0, mi)])
        z = zeta * (1.0 + trial)
        ref = lo.direction(g, z, S, Y, SS, L, D, Je=qp["Je"], Ji=qp["Ji"], s=s, lda=lda, reg=1e-12)
        dz, st = core.direction(g, s, lda, z, S, Y, SS, L, D, reg=1e-12)
        assert _rel(dz.cpu().numpy(), ref) <= 1e-9
        assert core.last_timings()["gram_launches"] == 1
    Je2 = qp["Je"] * 1.5
    core.stage_jacobian(Je2, qp["Ji"])
    ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=Je2, Ji=qp["Ji"], s=s, lda=lda, reg=1e-12)
    dz, st = core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=1e-12)
    assert _rel(dz.cpu().numpy(), ref) <= 1e-9
    assert core.last_timings()["gram_launches"] == 2
    core.close()


def test_ipm_lbfgs_linear_constraints_stage_once():
    """Problem 7 has linear constraints (README.md:60-75): with linear_constraints=True the Jacobians are staged
    once and the run is the same."""
    from pyipm_amd.ipm import IPM
    from pyipm_amd.problems import example_problem, unit_test_x0
    p = example_problem(7)
    runs = []
    for lin in (False, True):
        ipm = IPM(x0=unit_test_x0()[7], f=p["f"], df=p["df"], ce=p["ce"], dce=p["dce"], ci=p["ci"], dci=p["dci"],
                  lbfgs=4, Ftol=1.0e-8, verbosity=-1, device=0, linear_constraints=lin)
        x, s, lda, fval, kkt = ipm.solve()
        runs.append((x, ipm.iter_count, ipm.backend.n_staged, ipm.signal))
    assert runs[1][2] == 1 and runs[0][2] == runs[0][1]
    assert runs[0][1] == runs[1][1] and runs[0][3] == runs[1][3]
    np.testing.assert_allclose(runs[0][0], runs[1][0], rtol=1e-12, atol=1e-14)


def test_large_direction_satisfies_the_quasi_newton_system():
    """Size-independent property at a size the oracle cannot reach (its dense diag(1/A) alone would be 140 GB):
    the direction solves H dz = g for H = Z - U inv(M) U' (pyipm.py:1036-1052), applied matrix-free with torch as the
    checker.  n = 131072, p = 1024 (J = 1 GB), m = 8; also through the split-K Gram launch and the cached J'J."""
    import torch
    n, me, mi, m = 131072, 256, 768, 8
    p, N = me + mi, n + 2 * mi + me
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    J = torch.randn((n, p), generator=gen, dtype=torch.float64, device=dev) / np.sqrt(n)
    rng = np.random.default_rng(4)
    S = rng.standard_normal((n, m)) / np.sqrt(n)
    Mq = rng.standard_normal((n, 8)) / 3.0
    Y = Mq @ (Mq.T @ S) + 0.5 * S
    SY = S.T @ Y
    SS, L, D = S.T @ S, np.tril(SY, -1), np.diag(np.diag(SY))
    zeta = float(SY[-1, -1] / SS[-1, -1])
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    Sd, Yd = td(S), td(Y)
    core = _core(n, me, mi, m)
    core.stage_jacobian(J[:, :me], J[:, me:])
    for trial in range(2):                                  # second pass: other Sigma / zeta on the cached J'J
        s = td(rng.uniform(0.5, 2.0, mi))
        lda = td(np.concatenate([rng.standard_normal(me), rng.uniform(0.5, 2.0, mi)]))
        g = td(rng.standard_normal(N))
        z = zeta * (1.0 + trial)
        dz, st = core.direction(g, s, lda, z, Sd, Yd, SS, L, D, reg=1e-12)
        assert st["regularised"] == 0 and st["n_neg"] == 0 and st["n_zero"] == 0
        x, ds, dl = dz[:n], dz[n:n + mi], dz[n + mi:]
        W = torch.cat([z * Sd, Yd], dim=1)
        Minv = td(np.block([[z * SS, L], [L.T, -D]]))
        res = torch.empty_like(dz)
        res[:n] = z * x - W @ torch.linalg.solve(Minv, W.T @ x) + J @ dl
        res[n:n + mi] = lda[me:] / (s + EPS) * ds - dl[me:]
        low = J.t() @ x
        low[me:] -= ds
        res[n + mi:] = low
        assert float((res - g).norm() / g.norm()) <= 1e-10
    assert core.last_timings()["gram_launches"] == 1
    core.close()


# ---------------------------------------------------------------------- row-sharded direction (several ranks)
def _shard_problem(n, me, mi, m, seed):
    from pyipm_amd.problems import make_qp
    rng = np.random.default_rng(seed)
    qp = make_qp(n, me, mi, seed)
    zeta, S, Y, SS, L, D = _storage(n, m, rng, bool(me or mi), max(m, 1))
    g = rng.standard_normal(n + 2 * mi + me)
    s = qp["s"] if mi else np.zeros(0)
    lda = qp["lam"] if (me or mi) else np.zeros(0)
    return qp, zeta, S, Y, SS, L, D, g, s, lda


def _shard_worker(rank, world, port, shape, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyipm_amd.lbfgs import LbfgsCore
        n, me, mi, m, seed = shape
        qp, zeta, S, Y, SS, L, D, g, s, lda = _shard_problem(n, me, mi, m, seed)
        cut = [(n * r) // world for r in range(world + 1)]          # uneven row blocks
        a, b = cut[rank], cut[rank + 1]
        core = LbfgsCore(b - a, me, mi, max(m, 1), device=0, shard=True)
        core.stage_jacobian(qp["Je"][a:b] if me else None, qp["Ji"][a:b] if mi else None)
        res = []
        for trial in range(2):                                        # second direction reuses the reduced J'J
            gl = np.concatenate([g[a:b], g[n:]]) * (1.0 + trial)
            dz, st = core.direction(gl, s, lda, zeta, S[a:b], Y[a:b], SS, L, D, reg=1e-12)
            res.append(dz.cpu().numpy())
        torch.cuda.synchronize()
        out[rank] = (a, b, res, st, core.bytes_reduced, core.last_timings()["gram_launches"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (1000, 60, 140, 5, 1)), (3, (777, 0, 130, 4, 2)), (2, (640, 90, 0, 3, 3)),
                                         (3, (900, 0, 0, 6, 4)), (2, (500, 40, 80, 0, 5))])
def test_row_sharded_direction_on_one_gpu(world, shape):
    """The direction shards by rows of J with three sums over the ranks (pyipm_lbfgs_set_allreduce).  The box has one
    GPU, so the ranks share it and the sums ride a gloo group through the host — the kernels, the callback points and
    the buffers are those an RCCL group would use.  Every rank must return its rows of dz_x and the replicated
    dz_s / dz_lambda of the unsharded direction."""
    import torch.multiprocessing as mp
    from oracle import lbfgs_oracle as lo
    from test_gpu_dist import _free_port
    n, me, mi, m, seed = shape
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), shape, out), nprocs=world, join=True)
    qp, zeta, S, Y, SS, L, D, g, s, lda = _shard_problem(n, me, mi, m, seed)
    p_pad = ((me + mi + 127) // 128) * 128 if (me + mi) else 0
    r, rr = 2 * m, 2 * m + 1
    for trial in range(2):
        ref = lo.direction(g * (1.0 + trial), zeta, S, Y, SS, L, D, Je=qp["Je"] if me else None,
                           Ji=qp["Ji"] if mi else None, s=s, lda=lda, reg=1e-12)
        scale = np.linalg.norm(ref)
        for rk in range(world):
            a, b, res, st, nbytes, ngram = out[rk]
            dz = res[trial]
            assert np.linalg.norm(dz[:b - a] - ref[a:b]) <= 1e-9 * scale
            assert np.linalg.norm(dz[b - a:] - ref[n:]) <= 1e-9 * scale
            if rk:
                assert np.array_equal(dz[b - a:], out[0][2][trial][out[0][1] - out[0][0]:])     # replicated rows: bit for bit
    for rk in range(world):
        a, b, res, st, nbytes, ngram = out[rk]
        if me + mi:
            assert ngram == 1                                         # J'J reduced once, reused by the second direction
            assert nbytes == 8 * (p_pad * p_pad + 2 * (p_pad * rr + r * rr))
        else:
            assert nbytes == 8 * 2 * r


def test_singular_small_system_is_reported_and_raised():
    """Displacements orthogonal to their gradient changes (S'Y = 0) make the 2m x 2m system exactly singular: the
    C-ABI reports a zero pivot (stats, NaN direction, return code 0 — the host decides), the host backend raises like
    scipy's solve does in the reference (pyipm.py:18-20)."""
    from pyipm_amd.ipm import HipLbfgsBackend
    n, m = 50, 2
    rng = np.random.default_rng(0)
    S, Y = np.zeros((n, m)), np.zeros((n, m))
    S[0, 0] = S[1, 1] = 1.0
    Y[2, 0] = Y[3, 1] = 1.0
    SS, L, D = Y.T @ Y, np.triu(S.T @ Y), np.diag(np.diag(S.T @ Y))
    g = rng.standard_normal(n)
    core = _core(n, 0, 0, m)
    dz, st = core.direction(g, np.zeros(0), np.zeros(0), 1.0, S, Y, SS, L, D)
    assert st["small_pivot_min"] == 0.0 and not np.all(np.isfinite(dz.cpu().numpy()))
    core.close()
    be = HipLbfgsBackend(n, 0, 0, m, device=0)
    with pytest.raises(np.linalg.LinAlgError):
        be.lbfgs_direction(None, None, np.zeros(0), np.zeros(0), g, 1.0, S, Y, SS, L, D, 0.0, EPS)


def _ragged_lbfgs_shapes(count, seed):
    rng = np.random.default_rng(seed)
    edge = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    out = []
    for i in range(count):
        pick = lambda top: int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, top))   # noqa: E731
        n = pick(700) + 8
        me = 0 if rng.random() < 0.3 else min(pick(140), n - 1)
        mi = 0 if rng.random() < 0.3 else pick(300)
        m = int(rng.integers(0, 9))
        out.append((n, me, mi, m, 500 + i))
    return out


@pytest.mark.parametrize("n,me,mi,m,seed", _ragged_lbfgs_shapes(24, 77))
def test_ragged_lbfgs_shapes_vs_oracle(n, me, mi, m, seed):
    """Seeded ragged (n, me, mi, m): constraint counts on and off the 64 / 128 grid of the internal factorisation, empty
    blocks, empty storage, n barely above me — against the oracle."""
    from oracle import lbfgs_oracle as lo
    from pyipm_amd.problems import make_qp
    rng = np.random.default_rng(seed)
    qp = make_qp(n, me, mi, seed % 50)
    zeta, S, Y, SS, L, D = _storage(n, m, rng, bool(me or mi), max(m, 1))
    m_eff = S.shape[1]
    g = rng.standard_normal(n + 2 * mi + me)
    s = qp["s"] if mi else np.zeros(0)
    lda = qp["lam"] if (me or mi) else np.zeros(0)
    ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=qp["Je"] if me else None, Ji=qp["Ji"] if mi else None, s=s, lda=lda,
                       reg=1e-12)
    core = _core(n, me, mi, max(m_eff, 1), nb=128)
    core.stage_jacobian(qp["Je"] if me else None, qp["Ji"] if mi else None)
    dz, st = core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=1e-12)
    assert st["m"] == m_eff and st["regularised"] == 0
    # the constraint-space system is a Gram matrix: its condition number is the square of J's
    tol = 1e-9 if (me + mi) * 4 <= n else 1e-6
    assert _rel(dz.cpu().numpy(), ref) <= tol, st
    core.close()
