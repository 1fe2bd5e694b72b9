"""The factorisation on ARBITRARY symmetric matrices.  With me = mi = 0 the KKT matrix is the staged
matrix itself (pyipm.py:824-827: only its upper triangle is read), so any symmetric indefinite system can
be pushed through the C-ABI: inertia from the block pivots must equal the eigenvalue inertia (what
reghess counts, pyipm.py:1381) and the solve must match LAPACK's LU to ~cond*eps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(M, b, **kw):
    from pyipm_amd.newton import NewtonCore
    n = M.shape[0]
    core = NewtonCore(n, 0, 0, device=0, **kw)
    core.stage_blocks(np.triu(M))                       # junk-free upper triangle only: the lower half is never read
    core.stage_vectors(np.zeros(n))
    core.assemble(0.0, 0.0)
    st = core.factor()
    x = core.solve(rhs=b, flip=False).cpu().numpy()
    return x, st


def _inertia(M):
    w = np.linalg.eigvalsh(M)
    return int(np.sum(w < 0)), int(np.sum(w > 0))


@pytest.mark.parametrize("n", [1, 2, 3, 7, 31, 63, 64, 65, 100, 127, 128, 129, 200, 333, 640])
def test_random_indefinite(n):
    rng = np.random.default_rng(1000 + n)
    G = rng.standard_normal((n, n))
    M = 0.5 * (G + G.T) + np.diag(rng.standard_normal(n) * 2.0)        # GOE-like + diagonal: strongly indefinite
    b = rng.standard_normal(n)
    x, st = _solve(M, b)
    neg, pos = _inertia(M)
    assert st["n_zero"] == 0 and st["n_neg"] == neg and st["n_pos"] == pos
    ref = np.linalg.solve(M, b)
    cond = np.linalg.cond(M)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) <= max(1e-10, 50 * cond * np.finfo(float).eps)
    assert np.linalg.norm(M @ x - b) <= 1e-11 * (np.linalg.norm(M, 2) * np.linalg.norm(x) + np.linalg.norm(b))


@pytest.mark.parametrize("n", [2, 10, 40, 64])
def test_zero_diagonal_needs_2x2_pivots(n):
    rng = np.random.default_rng(7 + n)
    G = rng.standard_normal((n, n))
    M = 0.5 * (G + G.T)
    np.fill_diagonal(M, 0.0)
    b = rng.standard_normal(n)
    x, st = _solve(M, b)
    neg, pos = _inertia(M)
    assert st["n_2x2"] >= 1 and st["n_zero"] == 0 and (st["n_neg"], st["n_pos"]) == (neg, pos)
    assert np.linalg.norm(M @ x - b) <= 1e-11 * (np.linalg.norm(M, 2) * np.linalg.norm(x) + np.linalg.norm(b))


def test_saddle_point_block_matrix():
    """[[A, B'],[B, 0]] with A positive definite: exactly the structure of an equality-constrained KKT system."""
    rng = np.random.default_rng(5)
    n, m = 150, 60
    A = rng.standard_normal((n, n)); A = A @ A.T / n + np.eye(n)
    B = rng.standard_normal((m, n))
    M = np.block([[A, B.T], [B, np.zeros((m, m))]])
    b = rng.standard_normal(n + m)
    x, st = _solve(M, b)
    assert (st["n_neg"], st["n_pos"], st["n_zero"]) == (m, n, 0)
    assert np.linalg.norm(M @ x - b) <= 1e-11 * (np.linalg.norm(M, 2) * np.linalg.norm(x) + np.linalg.norm(b))


def test_badly_scaled_diagonal_blocks():
    """Late-IPM flavour: Sigma entries spanning 1e-8 .. 1e8 (diagonal), coupled by -I to a zero block."""
    rng = np.random.default_rng(9)
    k = 96
    sig = 10.0 ** rng.uniform(-8, 8, k)
    M = np.block([[np.diag(sig), -np.eye(k)], [-np.eye(k), np.zeros((k, k))]])
    b = rng.standard_normal(2 * k)
    x, st = _solve(M, b)
    assert st["n_zero"] == 0 and st["n_neg"] == k and st["n_pos"] == k
    ref = np.linalg.solve(M, b)
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) <= 1e-8        # cond ~1e16 * tiny coupling; LU gives the same class


def test_tile_local_pivoting_falls_back_to_static_pivots():
    """Pivots never leave their 64x64 diagonal tile.  A matrix whose leading tile is exactly singular although the
    matrix is not (here [[0, I],[I, 0]], 128 x 128) used to be a documented limitation (rejected pivots -> the host had
    to shift).  Now the zero pivots become static pivots (reported in n_zero, counted by their sign), the factor stays
    finite with the inertia of the matrix, and a plain solve is a sqrt(eps)-accurate preconditioner; the refined solve
    (tests/test_gpu_pivoting.py) recovers the exact answer.  A reghess-style shift still works as before."""
    k = 64
    M = np.block([[np.zeros((k, k)), np.eye(k)], [np.eye(k), np.zeros((k, k))]])
    x, st = _solve(M, np.ones(2 * k))
    assert st["n_zero"] == k and (st["n_neg"], st["n_pos"]) == _inertia(M) and st["nonfinite"] == 0
    assert np.isfinite(x).all() and np.linalg.norm(M @ x - 1.0) <= 1e-6 * np.sqrt(2 * k)
    from pyipm_amd.newton import NewtonCore
    core = NewtonCore(2 * k, 0, 0, device=0)
    core.stage_blocks(np.triu(M)); core.stage_vectors(np.zeros(2 * k))
    core.assemble(1e-3, 0.0)
    st2 = core.factor()
    neg, pos = _inertia(M + 1e-3 * np.eye(2 * k))
    assert st2["n_zero"] == 0 and (st2["n_neg"], st2["n_pos"]) == (neg, pos)


@pytest.mark.parametrize("shape,nb", [((1024, 256, 768, 1), 256), ((1000, 300, 900, 2), 256), ((2048, 0, 2048, 3), 256),
                                      ((1536, 512, 1024, 4), 128), ((700, 64, 1999, 5), 256)])
def test_structural_zero_skipping_is_bitwise_neutral(shape, nb):
    """The factorisation skips update tiles that the KKT block pattern leaves at exact zero and eliminates panels
    inside the slack block in closed form.  That must not change a single bit: directions
    and statistics equal those of the all-dense path, also with Sigma spread over 16 orders of magnitude (flagged
    tiles -> refined block solves, which the closed form replays as fused multiply-adds) and with block
    boundaries that do not sit on tile / panel boundaries."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    rng = np.random.default_rng(seed)
    sig = np.exp(rng.uniform(np.log(1e-8), np.log(1e8), mi))
    s = rng.uniform(0.5, 2.0, mi)
    lam = np.concatenate([qp["lam"][:me], sig * s])
    out = []
    for skip in (0, 1):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.set_option("skip_zeros", skip)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], s, lam, mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        g = core.residual()
        raw = core.solve(flip=False)
        out.append((dz.clone(), st, float((core.matvec(raw) - g).norm() / g.norm())))
        core.close()
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]
    assert out[0][1]["n_neg"] == me + mi and out[0][1]["n_zero"] == 0
    assert out[1][2] <= 1e-9


@pytest.mark.parametrize("shape,nb", [((1500, 300, 500, 11), 256), ((2048, 0, 2048, 3), 256), ((3000, 1000, 0, 7), 128),
                                      ((900, 64, 1800, 5), 256)])
def test_chain_kernel_choices_are_bitwise_neutral(shape, nb):
    """The updates that sit on the panel chain have two implementations (k_inpanel_update: 32 x 64 blocks straight from
    global memory; k_update: LDS-staged 128-wide tiles) chosen by how many rows remain, the panels of a group are chained tile to tile
    (group_chain, kernels_panel.hpp; the in-group updates of the rows below the group's diagonal block left-looking or panel
    by panel, pending_left_rows), or a panel is factored by the stepped schedule on its own (tile_step: one launch per
    diagonal tile + one for the rows below) or tile by tile
    with an inversion and a scaling launch over all rows (where the in-panel update of a tile can ride the scaling launch
    of the tile before it, fuse_scale_update), and the lookahead head can be applied panel by panel (early_head).  All of them accumulate the same products in the same order: every combination
    must give the same bits."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    out = []
    for opts in ({}, {"group_chain": 0}, {"group_chain": 0, "tile_step": 0}, {"tile_chain": 0}, {"tile_chain": 0, "group_chain": 0},
                 {"tail_group": 2}):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.set_option("expert", 1)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz, st = core.step(0.0, 0.0)
        out.append((dz.clone(), st, opts))
        core.close()
    base = out[0]
    for dz, st, opts in out[1:]:
        if "tail_group" in opts:
            continue                      # a different group schedule regroups the accumulation: equal only to rounding
        assert torch.equal(dz, base[0]), opts
        assert st == base[1], opts
    ref = out[-1][0]
    assert float((ref - base[0]).norm() / base[0].norm()) <= 1e-12


@pytest.mark.parametrize("shape,nb", [((1500, 300, 500, 11), 256), ((1000, 300, 900, 2), 256), ((2048, 0, 2048, 3), 256),
                                      ((3000, 1000, 0, 7), 128), ((700, 0, 0, 5), 128), ((900, 100, 650, 6), 128),
                                      ((3072, 768, 1152, 0), 256), ((130, 20, 30, 8), 256)])
def test_one_launch_backward_sweep_matches_the_per_panel_launches(shape, nb):
    """k_bwd_sweep (the whole backward substitution as one device-driven launch: workgroup 0 on the diagonal blocks, every
    other wave on its columns, flags and values handed over through agent-scope atomics) against the per-panel launches:
    equal to rounding, the same bits every time it runs (fixed ownership, fixed summation order), and it solves the system.
    Shapes: ragged last panels, no constraints, panels inside the slack block, a matrix of one panel (falls back)."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    core = NewtonCore(n, me, mi, device=0, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.set_option("sweep_persist", 0)
    ref, st0 = core.step(0.0, 0.0)
    ref = ref.clone()
    core.set_option("sweep_persist", 1)
    runs = []
    for _ in range(5):
        dz, st = core.step(0.0, 0.0)
        runs.append(dz.clone())
        assert st == st0
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    assert float((runs[0] - ref).norm() / ref.norm()) <= 1e-13
    g = core.residual()
    raw = core.solve(flip=False)
    assert float((core.matvec(raw) - g).norm() / g.norm()) <= 1e-12
    # a right-hand side of its own: the forward pass is not fused under a factorisation and runs as ONE launch too
    # (k_fwd_sweep + one k_diag_apply over all tiles); refinement runs both sweeps several times per solve
    rhs = torch.randn(core.N, dtype=torch.float64, device="cuda")
    x1 = core.solve(rhs, flip=False, refine=1)
    xa = core.solve(rhs, flip=False)
    xb = core.solve(rhs, flip=False)
    assert torch.equal(xa, xb)
    core.set_option("expert", 1)
    core.set_option("sweep_max_blocks", 5)               # few workgroups: every owner has several chunks / column groups
    xc = core.solve(rhs, flip=False)
    core.set_option("sweep_max_blocks", 0)
    core.set_option("sweep_persist", 0)
    x0 = core.solve(rhs, flip=False, refine=1)
    xd = core.solve(rhs, flip=False)
    assert float((x1 - x0).norm() / x0.norm()) <= 1e-12
    assert float((xa - xd).norm() / xd.norm()) <= 1e-13 and float((xc - xd).norm() / xd.norm()) <= 1e-13
    assert float((core.matvec(xa) - rhs).norm() / rhs.norm()) <= 1e-12
    core.close()


def test_one_launch_backward_sweep_with_nan_does_not_wait():
    """The workgroups of k_bwd_sweep recognise a value that has arrived by its not being NaN; a right-hand side that IS NaN
    must come back as NaN at once (the counts say the values are there), not after the 2 s poll timeout, and must not
    leave the handle in an error state."""
    import time
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi = 1500, 300, 500
    qp = make_qp(n, me, mi, 11)
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz, st = core.step(0.0, 0.0)
    good = dz.clone()
    rhs = torch.full((core.N,), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = core.solve(rhs, flip=False)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0
    assert bool(torch.isnan(x).all())
    rhs2 = torch.randn(core.N, dtype=torch.float64, device="cuda")
    rhs2[core.N // 2] = float("nan")                      # one NaN in the middle: everything it reaches is NaN, nothing hangs
    x = core.solve(rhs2, flip=False)
    torch.cuda.synchronize()
    assert bool(torch.isnan(x).any())
    dz2, st2 = core.step(0.0, 0.0)                       # factor_end would report a timed-out poll of an earlier sweep
    assert torch.equal(dz2, good) and st2 == st
    core.close()


def test_one_launch_sweeps_beside_another_streams_work():
    """The workgroups of k_fwd_sweep / k_bwd_sweep wait for each other, so they need the GPU to get all of them resident
    eventually: beside another stream that keeps every CU busy (large GEMMs here) the solve must still come back right --
    late, not wrong, and without tripping the 2 s poll timeout."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi = 3000, 700, 1200
    qp = make_qp(n, me, mi, 3)
    core = NewtonCore(n, me, mi, device=0)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core.step(0.0, 0.0)
    rhs = torch.randn(core.N, dtype=torch.float64, device="cuda")
    ref = core.solve(rhs, flip=False).clone()
    a = torch.randn(8192, 8192, dtype=torch.float64, device="cuda")
    other = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(other):
        for _ in range(12):
            a2 = a @ a                                   # ~14 ms each at fp64: every CU busy for the whole test
    outs = [core.solve(rhs, flip=False).clone() for _ in range(6)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    dz, st = core.step(0.0, 0.0)                         # (factor_end would report a timed-out poll)
    assert st["n_neg"] == me + mi
    core.close()


def _header_options():
    """(public, expert) option names as include/pyipm_newton.h documents them."""
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pyipm_newton.h")).read()
    pub = re.search(r"PUBLIC OPTIONS:(.*?)\n \*\n", txt, re.S).group(1)
    exp = re.search(r"EXPERT OPTIONS:(.*?)\n \*   \(which stream", txt, re.S).group(1)
    names = lambda blk: [w for w in re.sub(r"[*\n]", " ", blk).replace(",", " ").split() if w]     # noqa: E731
    return names(pub), names(exp)


# options that legitimately change the numbers (another system form, another rounding order, tolerances) or are not schedule
# choices at all -- everything else in the header must be in the bitwise-neutrality sweep below
NOT_SCHEDULE = {"expert", "condensed", "condensed_sigma_max", "condensed_refine", "block_refine", "refine_cond", "refine_target",
                "refine_max", "pivtol_rel", "tile_blocked", "profile", "sweep_persist",
                # multi-rank / per-panel schedule: swept in tests/test_gpu_dist.py (exchange forms, wide panels, slices, the time bound)
                "wide_sub", "dist_sag", "dist_sag_min_bytes", "dist_slices", "dist_selfmsg", "dist_comm2", "dist_timeout_s",
                # test hooks and diagnostics
                "bc_per_problem",                         # (batched handles: swept in tests/test_gpu_batched.py)
                "sweep_max_blocks", "debug_fault", "debug_timeline_ptr", "debug_chain_ptr"}
SCHEDULE_SPACE = {"lookahead": [0, 1, 2], "group": [1, 2, 4, 8], "group_chain": [0, 1], "fuse_forward": [0, 1], "keep_zeros": [0, 1],
                  "skip_zeros": [0, 1], "tile_step": [0, 1],
                  # round 3: 128 x 256 bulk tiles, persistent bulk launches that leave CUs to the chain
                  "bulk_bn": [128, 256], "reserve_cus": [0, 16, 64], "persist_rows": [0, 4096, 1 << 20],
                  "tail_group": [2, 4, 8],
                  # round 5: the tile steps' critical block on eight waves (chain + helpers)
                  "tile_waves": [4, 8, 9],
                  # round 6: the tile steps of a diagonal block as one launch of persistent workgroups
                  "tile_chain": [0, 1, 2], "chain_cpy": [2, 5, 9], "chain_whole": [0, 1], "chain_lds_kb": [0, 100]}


def test_option_lists_header_library_and_sweep_agree():
    """include/pyipm_newton.h documents 28 public options and gates the 12 others behind PYIPM_EXPERT / set_option("expert", 1)
    (at most 40 names, at most 15 of them expert).  The header's two lists, what the library accepts, and what the bitwise-neutrality sweep covers are
    the same names: an undocumented, ungated or unswept option cannot exist."""
    import os
    from pyipm_amd.newton import NewtonCore, NewtonError
    pub, exp = _header_options()
    assert len(pub) == len(set(pub)) and len(exp) == len(set(exp)) <= 15 and len(pub) + len(exp) <= 40 and not set(pub) & set(exp)
    assert set(SCHEDULE_SPACE) | NOT_SCHEDULE == set(pub) | set(exp), (set(SCHEDULE_SPACE) | NOT_SCHEDULE) ^ (set(pub) | set(exp))
    assert not set(SCHEDULE_SPACE) & NOT_SCHEDULE
    saved = os.environ.pop("PYIPM_EXPERT", None)
    try:
        core = NewtonCore(200, 30, 50, device=0)
        with pytest.raises(NewtonError, match="unknown option"):
            core.set_option("no_such_option", 1)
        for name in exp:                                  # refused without the gate ...
            with pytest.raises(NewtonError, match="expert switch"):
                core.set_option(name, 0)
        for name in pub:                                  # ... public ones accepted (with a harmless value)
            core.set_option(name, {"group": 1, "bulk_bn": 256, "wide_sub": 256, "refine_max": 8, "refine_target": 1e-14,
                                   "refine_cond": 1e3, "pivtol_rel": 1e-14, "condensed_sigma_max": 1e4, "block_refine": 2,
                                   "persist_rows": 12288, "reserve_cus": 16, "dist_sag_min_bytes": 4 << 20, "dist_timeout_s": 300.0,
                                   "tile_chain": 1}.get(name, 0))
        core.set_option("expert", 1)                      # ... and the handle-level gate opens all of them
        for name in exp:
            core.set_option(name, 0)
        core.close()
    finally:
        if saved is not None:
            os.environ["PYIPM_EXPERT"] = saved


def test_random_schedule_options_give_the_same_bits():
    """The schedule of a factorisation -- which stream runs what, in how many launches, how much is looked ahead, what is
    skipped as structurally zero, whether zeros are left in place between assemblies -- must never show in the result.
    Random combinations of the scheduling options, three steps in a row on one handle each (stale state and races show up
    as run-to-run differences): every direction equals the default configuration's bit for bit."""
    import random
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    rnd = random.Random(7)
    space = SCHEDULE_SPACE
    for shape, nb in (((3000, 700, 1200, 3), 256), ((1900, 300, 900, 9), 128), ((5000, 1000, 2500, 11), 256),
                      ((1000, 300, 900, 2), 256)):       # (the last one: a lone 128-wide panel as last group, n off every tile boundary)
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        ref = None
        for trial in range(10):
            opts = {} if trial == 0 else {k: rnd.choice(v) for k, v in space.items() if rnd.random() < 0.4}
            core = NewtonCore(n, me, mi, device=0, nb=nb)
            core.set_option("expert", 1)                     # (the only place the suite needs every switch: no global PYIPM_EXPERT)
            # (the substitution sweeps as per-panel launches throughout: the one-launch sweeps sum in another order, so with
            # them fuse_forward -- forward pass under the factorisation, panel by panel, or after it, in one launch -- would
            # show in the last bits; they have their own test above)
            core.set_option("sweep_persist", 0)
            for k, v in opts.items():
                core.set_option(k, v)
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
            core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
            for rep in range(3):
                dz, st = core.step(0.0, 0.0)
                if ref is None:
                    ref = dz.clone()
                    raw = dz.clone(); raw[n + mi:] *= -1.0            # ... and the default is RIGHT: |Hc dz - g| / |g| from the blocks
                    gres = core.residual()
                    assert float((core.matvec(raw) - gres).norm() / gres.norm()) <= 1e-12, (shape, nb)
                assert torch.equal(dz, ref), (shape, nb, opts, rep)
            core.close()


@pytest.mark.parametrize("shape", [(1900, 300, 500, 5), (2400, 0, 0, 6), (1000, 300, 900, 2)])
def test_wide_bulk_tiles_stop_at_the_storage_edge(shape):
    """Npad is a multiple of 128 only, a 128 x 256 bulk tile at the last 128 columns of such a matrix
    would read and rewrite 128 columns PAST the storage -- the first W slot, which the group after next reuses while the
    launch runs.  Shapes with Npad % 256 == 128, wide tiles forced everywhere (no reserved CUs, no row threshold), groups
    of one and two panels so that every W slot is reused several times; the dense (me = mi = 0) shape has no structural
    zeros that could hide a clobbered W.  Bit for bit the 128 x 128 schedule, three steps in a row, and nothing outside the
    lower triangle's storage changes (the W buffer is compared through the next factorisation's bits)."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    outs = {}
    for bn in (128, 256):
        for grp in (1, 2):
            core = NewtonCore(n, me, mi, device=0, nb=256)
            assert core.Npad % 256 == 128
            for k, v in (("expert", 1), ("bulk_bn", bn), ("reserve_cus", 0), ("group", grp),
                         ("tail_group", grp), ("sweep_persist", 0)):
                core.set_option(k, v)
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
            core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
            steps = [core.step(0.0, 0.0)[0].clone() for _ in range(3)]
            assert all(torch.equal(s, steps[0]) for s in steps), (shape, bn, grp)
            outs[(bn, grp)] = steps[0]
            core.close()
    for grp in (1, 2):
        assert torch.equal(outs[(128, grp)], outs[(256, grp)]), (shape, grp)


def _poison(core):
    """Every byte of the handle's workspace that the library has not initialised itself becomes a NaN pattern: a kernel that
    reads memory nobody wrote (a fresh process hands out zero pages, which hides it) turns the direction into NaN."""
    import torch
    ws = core.workspace.view(torch.float64)
    keep = ws.clone()
    ws.fill_(float("nan"))
    return keep


@pytest.mark.parametrize("shape,nb", [((1408, 300, 768, 3), 256), ((1280, 256, 896, 4), 256), ((2176, 0, 1472, 5), 256), ((1500, 356, 700, 6), 256),
                                      ((1408, 300, 768, 3), 128)])
def test_no_kernel_reads_memory_nobody_wrote(shape, nb):
    """Round 4: with the slack rows of x-block panels skipped in whole 128-row tiles, a 128 x 256 bulk tile whose columns
    straddle the edge of that hole read W rows that no kernel had written -- stale memory (zero pages in a fresh process: every
    test passed).  Shapes with n and / or n + mi an odd multiple of 128, wide tiles forced everywhere, the whole workspace
    poisoned with NaN before anything is staged: the direction is finite, meets the blocks to 1e-12 and has the bits of the
    128 x 128 schedule -- for several steps on one handle (the W slots rotate)."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.problems import make_qp
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    outs = {}
    for bn in (256, 128):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        _poison(core)
        for k, v in (("expert", 1), ("bulk_bn", bn), ("reserve_cus", 0), ("group", 2), ("tail_group", 2),
                     ("sweep_persist", 0)):
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        steps = [core.step(0.0, 0.0)[0].clone() for _ in range(3)]
        assert all(bool(torch.isfinite(s).all()) for s in steps), (shape, bn)
        assert all(torch.equal(s, steps[0]) for s in steps), (shape, bn)
        raw = steps[0].clone(); raw[n + mi:] *= -1.0
        gres = core.residual()
        assert float((core.matvec(raw) - gres).norm() / gres.norm()) <= 1e-12, (shape, bn)
        outs[bn] = steps[0]
        core.close()
    assert torch.equal(outs[128], outs[256]), shape
