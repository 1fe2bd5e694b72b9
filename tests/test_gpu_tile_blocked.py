"""The 64x64 tile inversion in blocks of 16 pivots (csrc/tile_blocked.hpp) against NumPy and against the single sweeps it
falls back to.  A KKT system with n <= 64 and no constraints IS one diagonal tile, so step() = tile inversion + one
product: every family below goes through the fast path, leaves it at a chosen micro-block, or never enters it.
Replaces the LAPACK factorisation reached from pyipm.py:18-20, 1720 (the reference solves these by LU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(7)


def _spd(n, cond=None):
    M = RNG.standard_normal((n, n))
    if cond is None:
        return M @ M.T / n + np.eye(n)
    Q, _ = np.linalg.qr(M)
    return (Q * np.logspace(0, np.log10(cond), n)) @ Q.T


def _quasi(n, k):
    H = _spd(k)
    D = np.diag(RNG.uniform(0.5, 2.0, n - k))
    J = RNG.standard_normal((k, n - k)) / np.sqrt(k)
    return np.block([[H, J], [J.T, -D]])


def _families():
    yield "spd", _spd(64), 4
    yield "spd padded n=48", _spd(48), 4
    yield "spd n=17", _spd(17), 4
    yield "negative definite", -_spd(64), 4
    yield "quasi-definite 40+24", _quasi(64, 40), 4
    yield "quasi-definite 16+48", _quasi(64, 16), 4
    yield "spd cond 1e8", _spd(64, 1e8), None          # (graded: natural-order multipliers exceed 1/alpha somewhere -- Bunch-Kaufman's call)
    S = np.diag(np.logspace(-8, 8, 64))
    E = RNG.standard_normal((64, 64)) * 1e-9
    yield "diag 1e-8..1e8 + tiny dense", S + (E + E.T), 4
    Z = RNG.standard_normal((64, 64)); Z = Z + Z.T; np.fill_diagonal(Z, 0.0)
    yield "zero diagonal: 2x2 pivots from the first micro-block", Z, 0
    for blk in (1, 2, 3):                                   # the fast path commits `blk` micro-blocks, then hands over
        A = _spd(64)
        k = 16 * blk + 5
        A[k, k] = 0.0
        A[k, :k] *= 1e-3; A[:k, k] *= 1e-3
        yield "spd with a zero diagonal entry at %d" % k, A, blk
    B = _quasi(64, 32); B[50, 50] = 1e-30
    yield "quasi-definite with a tiny diagonal entry at 50", B, 4      # (its Schur complement is regular: -J'inv(H)J)
    G = RNG.standard_normal((64, 64)); G = G + G.T
    yield "random symmetric indefinite", G, None


@pytest.mark.parametrize("name,H,blocks_expected", list(_families()), ids=[f[0] for f in _families()])
def test_tile_inversion_blocked_and_single_sweeps(name, H, blocks_expected):
    import torch
    from pyipm_amd.newton import NewtonCore
    n = H.shape[0]
    g = RNG.standard_normal(n)
    ref = np.linalg.solve(H, g)
    w = np.linalg.eigvalsh(H)
    cond = np.abs(w).max() / np.abs(w).min()
    out = {}
    for blocked in (1, 0):
        core = NewtonCore(n, 0, 0, device=0)
        core.set_option("expert", 1)
        core.set_option("tile_blocked", blocked)
        buf = torch.zeros(256, dtype=torch.int64, device="cuda")
        core.stage_blocks(np.triu(H) + np.triu(H, 1).T, None, None)
        core.stage_vectors(-g, None, None, None, np.zeros(0), mu=0.1)            # residual = -df = g
        core.set_option("debug_timeline_ptr", float(buf.data_ptr()))
        dz, st = core.step(0.0, 0.0)
        torch.cuda.synchronize()
        core.set_option("debug_timeline_ptr", 0.0)
        dz = dz.cpu().numpy()
        out[blocked] = (dz, st, buf.cpu().numpy())
        assert st["n_neg"] == int((w < 0).sum()) and st["nonfinite"] == 0, (name, blocked, st)
        assert np.linalg.norm(H @ dz - g) <= 1e-13 * cond * np.linalg.norm(g) + 1e-14 * np.linalg.norm(g), (name, blocked)
        assert np.linalg.norm(dz - ref) <= 20 * np.finfo(float).eps * cond * np.linalg.norm(ref), (name, blocked)
    (dz1, st1, b1), (dz0, st0, _) = out[1], out[0]
    assert (st1["n_neg"], st1["n_zero"], st1["n_2x2"]) == (st0["n_neg"], st0["n_zero"], st0["n_2x2"])
    if blocks_expected is not None:
        # diagnostics of the fast path: dbg[7] = micro-blocks committed, summed over the tiles of the factorisation -- the
        # tile under test and the identity tile that pads N = 64 to the 128-row storage (4 blocks)
        committed = int(b1[7]) - 4
        assert committed == blocks_expected, (name, committed, blocks_expected)
    if blocks_expected == 0:
        np.testing.assert_array_equal(dz1, dz0)             # nothing committed: the same sweeps from the same state


def _graded_qp(n, me, mi, seed, decades):
    """A convex QP whose x-x tiles have pivot spreads far beyond refine_cond (block solves with them are refined) and, with
    `decades` large enough, micro-blocks the natural-order check rejects (the tile inversion hands over to the single sweeps
    in the middle of a tile)."""
    from pyipm_amd.problems import make_qp
    qp = make_qp(n, me, mi, seed)
    rng = np.random.default_rng(seed)
    sc = np.logspace(0.0, decades, n)[rng.permutation(n)]
    qp["d2L"] = qp["d2L"] * np.sqrt(sc)[:, None] * np.sqrt(sc)[None, :]
    return qp


@pytest.mark.parametrize("shape", [(700, 100, 200, 1, 5.0), (1100, 0, 300, 2, 9.0), (520, 130, 0, 3, 3.0), (2100, 300, 500, 4, 6.0)])
def test_eight_wave_tile_step_gives_the_bits_of_the_four_wave_one(shape):
    """k_tile_step8 (chain waves + helper waves, csrc/kernels_panel.hpp, tile_blocked8.hpp) against k_tile_step on systems
    whose tiles exercise every branch the helpers have to keep in step with: refined block solves (flagged tiles: four
    barriers per refinement step that the helper waves mirror), micro-blocks that fail the Bunch-Kaufman check after one to
    three committed blocks (the helpers leave, the critical waves go on with the single sweeps), units paired two per
    block or one per block.  Factor storage, statistics and direction are bit for bit the same; three steps in a row per
    handle.  Replaces the LAPACK factorisation reached from pyipm.py:18-20, 1720-1721."""
    import torch
    from pyipm_amd.newton import NewtonCore
    n, me, mi, seed, decades = shape
    qp = _graded_qp(n, me, mi, seed, decades)
    out = {}
    for key, opts in (("w4", {"tile_waves": 4}),
                      ("w8", {"tile_waves": 9}),
                      ("w8_exposed", {"tile_waves": 8}),
                      ("w8_single_sweeps", {"tile_waves": 9, "tile_blocked": 0}),
                      ("w4_single_sweeps", {"tile_waves": 4, "tile_blocked": 0})):
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("expert", 1)
        core.set_option("sweep_persist", 0)
        core.set_option("tile_chain", 0)                  # (the launch-per-tile kernels are what is compared here)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        for rep in range(3):
            dz, st = core.step(0.0, 0.0)
            fac = core.kkt_storage().clone().view(torch.int64)      # (bit patterns: what no kernel writes is NaN-poisoned by the suite)
            if key not in out:
                out[key] = (dz.clone(), fac, st)
            assert torch.equal(dz, out[key][0]) and torch.equal(fac, out[key][1]), (shape, key, rep)
        core.close()
    for a, b in (("w4", "w8"), ("w4", "w8_exposed"), ("w4_single_sweeps", "w8_single_sweeps")):
        assert torch.equal(out[a][1], out[b][1]), (shape, a, b, "factor storage")
        assert torch.equal(out[a][0], out[b][0]), (shape, a, b, "direction")
        sa, sb = out[a][2], out[b][2]
        assert {k: sa[k] for k in sa if not k.endswith("_ms")} == {k: sb[k] for k in sb if not k.endswith("_ms")}, (shape, a, b)
    st = out["w4"][2]
    assert st["n_neg"] == me + mi and st["nonfinite"] == 0, (shape, st)


@pytest.mark.parametrize("shape", [(700, 100, 200, 1, 5.0), (1100, 0, 300, 2, 9.0), (520, 130, 0, 3, 3.0), (2100, 300, 500, 4, 6.0),
                                   (2048, 0, 2048, 5, 0.0), (4100, 1000, 700, 6, 4.0)])
def test_tile_chain_in_one_launch_gives_the_bits_of_the_launch_per_tile(shape):
    """k_tile_chain (csrc/kernels_chain.hpp, round 6): the tile steps of a diagonal block as ONE launch of persistent
    workgroups -- the chain workgroup and units that own row tiles, handing inv(T), W and the diagonal tiles over through
    written-through stores and one progress word per workgroup -- against one launch per tile (k_tile_step, four and eight
    waves).  Graded systems: refined block solves, micro-blocks that fail the Bunch-Kaufman check in the middle of a tile
    (single sweeps, 2 x 2 pivots), several groups, ragged last panels; units per row tile 1 .. 4 (chain_cpy).  Factor storage,
    statistics and direction are bit for bit the same; three steps in a row per handle (the words' epochs), in the single-rank
    schedule and in the per-panel one (factor_panel: a chain of 4 tiles per panel).  Replaces the LAPACK factorisation
    reached from pyipm.py:18-20, 1720-1721."""
    import torch
    from pyipm_amd.newton import NewtonCore
    n, me, mi, seed, decades = shape
    qp = _graded_qp(n, me, mi, seed, decades)
    out = {}
    for key, opts in (("steps4", {"tile_chain": 0, "tile_waves": 4}),
                      ("steps8", {"tile_chain": 0, "tile_waves": 9}),
                      ("chain", {"tile_chain": 2}),
                      ("chain_exposed", {"tile_chain": 1}),
                      ("chain_pieces", {"tile_chain": 2, "chain_whole": 0}),
                      ("chain_cpy2", {"tile_chain": 2, "chain_cpy": 2}),
                      ("chain_cpy9", {"tile_chain": 2, "chain_cpy": 9}),
                      ("chain_group2", {"tile_chain": 2, "group": 2, "tail_group": 2}),
                      ("steps_single", {"tile_chain": 0, "tile_waves": 4, "tile_blocked": 0}),
                      ("chain_single", {"tile_chain": 2, "tile_blocked": 0})):
        core = NewtonCore(n, me, mi, device=0)
        core.set_option("expert", 1)
        core.set_option("sweep_persist", 0)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        for rep in range(3):
            dz, st = core.step(0.0, 0.0)
            fac = core.kkt_storage().clone().view(torch.int64)
            if key not in out:
                out[key] = (dz.clone(), fac, st)
            assert torch.equal(dz, out[key][0]) and torch.equal(fac, out[key][1]), (shape, key, rep)
        core.close()
    for a, b in (("steps4", "steps8"), ("steps4", "chain"), ("steps4", "chain_exposed"), ("steps4", "chain_pieces"), ("steps4", "chain_cpy2"), ("steps4", "chain_cpy9"),
                 ("steps4", "chain_group2"), ("steps_single", "chain_single")):
        assert torch.equal(out[a][1], out[b][1]), (shape, a, b, "factor storage")
        assert torch.equal(out[a][0], out[b][0]), (shape, a, b, "direction")
        sa, sb = out[a][2], out[b][2]
        assert {k: sa[k] for k in sa if not k.endswith("_ms")} == {k: sb[k] for k in sb if not k.endswith("_ms")}, (shape, a, b)
    st = out["steps4"][2]
    assert st["n_neg"] == me + mi and st["nonfinite"] == 0, (shape, st)


def test_tile_chain_in_the_per_panel_schedule():
    """The per-panel (multi-GPU) schedule on one rank: a chain of nb / 64 tiles per panel as one launch, nb = 256 and the wide
    panels of nb = 1024 (factor_wide_panel), against the launch-per-tile form."""
    import torch
    from pyipm_amd.newton import NewtonCore
    n, me, mi = 1500, 200, 400
    qp = _graded_qp(n, me, mi, 11, 4.0)
    for nb in (256, 1024):
        out = {}
        for key, opts in (("steps", {"tile_chain": 0}), ("chain", {"tile_chain": 1})):
            core = NewtonCore(n, me, mi, device=0, nb=nb)
            core.set_option("expert", 1)
            core.set_option("sweep_persist", 0)
            for k, v in opts.items():
                core.set_option(k, v)
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
            core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
            core.assemble(0.0, 0.0)
            core.factor_begin()
            for p in range(core.npanels):
                core.factor_panel(p)
                core.trailing_update(p)
            st = core.factor_end()
            fac = core.kkt_storage().clone().view(torch.int64)
            out[key] = (fac, st)
            core.close()
        assert torch.equal(out["steps"][0], out["chain"][0]), nb
        sa, sb = out["steps"][1], out["chain"][1]
        assert {k: sa[k] for k in sa if not k.endswith("_ms")} == {k: sb[k] for k in sb if not k.endswith("_ms")}, nb


@pytest.mark.parametrize("shape", [(2048, 0, 2048, 256, 8, 5, 3.0), (512, 64, 640, 128, 4, 6, 5.0), (1024, 100, 1100, 256, 4, 7, 2.0)])
def test_first_group_panel_by_panel_gives_the_bits_of_the_bulk_update(shape):
    """lookahead = 2 (factor_all, round 6): where the x block is ONE group and the slack block follows it (config 2), every panel of
    that group is applied to the columns beyond it as soon as it is complete -- K = nb launches on the side stream under the
    group's own chain, 128 x 64 tiles where a launch has few -- instead of one bulk update that the multiplier block's chain
    waits for.  Against the one-group lookahead and no lookahead at all: factor storage, statistics and direction bit for bit
    the same, three steps per handle; and the pieces did run (more update launches than with the bulk update).  Replaces the
    LAPACK factorisation reached from pyipm.py:18-20, 1720-1721."""
    import torch
    from pyipm_amd.newton import NewtonCore
    n, me, mi, nb, group, seed, decades = shape
    qp = _graded_qp(n, me, mi, seed, decades)
    out, launches = {}, {}
    for key, opts in (("bulk", {"lookahead": 1}), ("pieces", {"lookahead": 2}), ("none", {"lookahead": 0}),
                      ("pieces_steps", {"lookahead": 2, "tile_chain": 0}), ("pieces_bn128", {"lookahead": 2, "bulk_bn": 128})):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.set_option("expert", 1)
        core.set_option("profile", 1)
        core.set_option("group", group)
        for k, v in opts.items():
            core.set_option(k, v)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        for rep in range(3):
            dz, st = core.step(0.0, 0.0)
            fac = core.kkt_storage().clone().view(torch.int64)
            if key not in out:
                out[key] = (dz.clone(), fac, st)
            assert torch.equal(dz, out[key][0]) and torch.equal(fac, out[key][1]), (shape, key, rep)
        launches[key] = sum(v["launches"] for v in core.trailing_instances().values())
        core.close()
    for b in ("pieces", "none", "pieces_steps", "pieces_bn128"):
        assert torch.equal(out["bulk"][1], out[b][1]), (shape, b, "factor storage")
        assert torch.equal(out["bulk"][0], out[b][0]), (shape, b, "direction")
        sa, sb = out["bulk"][2], out[b][2]
        assert {k: sa[k] for k in sa if not k.endswith("_ms")} == {k: sb[k] for k in sb if not k.endswith("_ms")}, (shape, b)
    if n >= 2048:          # (launches of less than 30 us are not counted: factor_end takes them for empty tile lists)
        assert launches["pieces"] > launches["bulk"], (shape, launches)
    st = out["bulk"][2]
    assert st["n_neg"] == me + mi and st["nonfinite"] == 0, (shape, st)
