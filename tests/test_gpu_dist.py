"""GPU tests of the distributed HIP path.  The GPU box has ONE device, so world_size-2/3 runs put every rank on
cuda:0 and ride a gloo group with host staging (RCCL refuses duplicate devices).  What runs is what the multi-GPU
bench runs: the library's own per-panel driver (pyipm_newton_step_dist: lookahead schedule, pack / unpack messages,
the non-owner rebuild of L, segment sums and broadcasts of the sweeps) with the exchange bound to callbacks, the
row-sharded staging, refinement across the ranks -- and, for comparison, the Python loop over the per-panel phases.
A handle-owned RCCL communicator is exercised for real with a world of one rank."""
import os
import socket

import numpy as np
import pytest

from oracle import newton_oracle as orc
from pyipm_amd.problems import make_qp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shape, nb, mode, out, opts=None, drv_kw=None, env=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for k_, v_ in (env or {}).get(rank, {}).items():          # per-rank environment (PYIPM_DIST_SAG on ONE rank, ...)
        os.environ[k_] = v_
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyipm_amd.newton import NewtonCore
        from pyipm_amd.dist import DistNewton
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        core = NewtonCore(n, me, mi, device=0, nb=nb, world=world, rank=rank)
        for k, v in (opts or {}).items():
            core.set_option(k, v)
        if opts:
            core.set_option("profile", 1)                # (per-instance launch counts below)
        if mode == "native-sharded":                     # a rank stages only the rows of the x-columns it owns
            rows = core.owned_rows()
            core.stage_blocks_owned(qp["d2L"][rows], qp["Je"][rows] if me else None, qp["Ji"][rows] if mi else None)
        else:
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        drv = DistNewton(core, native=mode.startswith("native"), **(drv_kw or {}))
        drv.lookahead = mode != "python-lockstep"
        g = drv.residual().cpu().numpy()
        dz, st = drv.step(0.0, 0.0)
        extra = {}
        if mode.startswith("native"):
            raw = dz.clone()
            raw[n + mi:] *= -1.0
            y = core.matvec_dist(raw)                     # Hc dz from the ranks' blocks, summed over the ranks
            extra["berr"] = float((y.cpu() - torch.from_numpy(g)).norm() / np.linalg.norm(g))
            dz_ref = core.solve_dist(flip=True, refine=-1)
            extra["info"] = core.solve_info()
            extra["refined_diff"] = float((dz_ref - dz).norm() / dz.norm())
            extra["timings"] = core.dist_timings()
            extra["wire"] = core.dist_wire()                # (of the last factorisation: the step above)
            extra["bcast_mode"] = core.comm_bcast_mode()
            if opts:
                core.step_dist(0.0, 0.0)
                extra["instances"] = core.trailing_instances()
        torch.cuda.synchronize()
        out[rank] = (dz.cpu().numpy(), st, core.ncols_local, drv.bytes_broadcast, g, extra)
    finally:
        dist.destroy_process_group()


def _factor_bytes(n, me, mi, nb):
    """One message per panel that has rows below it, EXCEPT panels inside the slack block (every rank derives their
    contribution from s / lambda locally); a panel inside the x block leaves the whole 256-row blocks of slack rows home (256:
    the widest column tile of an update launch -- a 128 x 256 tile must never meet a W row nobody wrote, round 4)."""
    N = n + 2 * mi + me
    Npad = ((N + 127) // 128) * 128
    fact = 0
    for p in range((Npad + nb - 1) // nb):
        c0 = p * nb
        w = min(nb, Npad - c0)
        m = Npad - (c0 + w)
        in_s = c0 >= n and c0 + w <= n + mi
        if c0 + w <= n and mi:
            m -= max(0, (n + mi) // 256 * 256 - (n + 255) // 256 * 256)
        if m > 0 and not in_s:
            fact += 8 * (m * w + 2 * (w // 64) * 4096 + w // 64)
    return fact, Npad


@pytest.mark.parametrize("mode", ["native", "native-sharded", "python-lockstep", "python-lookahead"])
@pytest.mark.parametrize("world,shape,nb", [(2, (300, 100, 150, 7), 128), (2, (900, 200, 300, 8), 256),
                                            (3, (700, 150, 260, 9), 128), (2, (1400, 0, 400, 10), 128),
                                            # panels wider than 256 columns: the owner factors them as a block of 256-column
                                            # sub-panels (factor_wide_panel), the sweeps run sub-panel by sub-panel
                                            (2, (1300, 300, 450, 11), 512), (3, (2000, 400, 600, 12), 1024)])
def test_ranks_sharing_one_gpu(world, shape, nb, mode):
    import torch.multiprocessing as mp
    n, me, mi, seed = shape
    N = n + 2 * mi + me
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, nb, mode, out), nprocs=world, join=True)
    qp = make_qp(n, me, mi, seed)
    ref, _, _, gref = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                      qp["mu"], n, me, mi, regularise=False)
    cols = 0
    for r in range(world):
        dz, st, ncl, _, g, extra = out[r]
        np.testing.assert_allclose(g, gref, rtol=0, atol=1e-13 * np.abs(gref).max())      # the ranks' shares, summed
        assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= 1e-10
        assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == N - me - mi
        cols += ncl
        if extra:
            assert extra["berr"] <= 1e-12
            assert extra["info"]["backward_error"] <= 1e-13 and extra["refined_diff"] <= 1e-10
    fact, Npad = _factor_bytes(n, me, mi, nb)
    assert cols == Npad
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0])      # every rank ends with the same direction, bit for bit
    if mode.startswith("native"):
        assert out[0][5]["timings"]["bytes"] == fact     # bytes of the factorisation's panel messages
    else:
        # the Python driver counts the sweeps too: nb numbers summed per panel forward, nb broadcast backward
        assert out[0][3] == fact + 16 * Npad


def _ragged_dist_cases():
    """Ranks, panel widths and block boundaries that do not line up (a last panel narrower than nb, ranks without a panel in some
    block, empty blocks); PYIPM_DIST_FUZZ=count[,seed] adds seeded random cases by hand."""
    out = [(2, (129, 0, 65, 21), 128), (3, (515, 129, 0, 22), 256), (4, (200, 70, 333, 23), 128), (4, (1100, 1, 127, 24), 512)]
    spec = os.environ.get("PYIPM_DIST_FUZZ", "")
    if spec:
        count, seed = (spec.split(",") + ["4"])[:2]
        rng = np.random.default_rng(int(seed))
        for i in range(int(count)):
            n = int(rng.integers(2, 1500))
            me = 0 if rng.random() < 0.3 else int(rng.integers(1, min(n, 500) + 1))
            mi = 0 if rng.random() < 0.3 else int(rng.integers(1, 700))
            out.append((int(rng.integers(2, 5)), (n, me, mi, 5000 + i), int(rng.choice([128, 256, 512, 1024]))))
    return out


@pytest.mark.parametrize("world,shape,nb", _ragged_dist_cases())
def test_ranks_sharing_one_gpu_ragged(world, shape, nb):
    import torch.multiprocessing as mp
    n, me, mi, seed = shape
    N = n + 2 * mi + me
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, nb, "native-sharded", out), nprocs=world, join=True)
    qp = make_qp(n, me, mi, seed)
    ref, _, Hc, gref = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                       qp["mu"], n, me, mi, regularise=False)
    tol = max(1e-10, 20 * np.linalg.cond(Hc) * np.finfo(float).eps)
    for r in range(world):
        dz, st, ncl, _, g, extra = out[r]
        np.testing.assert_allclose(g, gref, rtol=0, atol=1e-13 * np.abs(gref).max())
        assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= tol
        assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == N - me - mi
        assert extra["berr"] <= 1e-12
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0])
    assert out[0][5]["timings"]["bytes"] == _factor_bytes(n, me, mi, nb)[0]


# (Npad a multiple of 256: 2560 and 8192 -- with an odd multiple of 128 the wide tiles are not used at all; n and
#  n + mi odd multiples of 128 in the second shape: a wide tile straddles both edges of the slack rows)
@pytest.mark.parametrize("world,shape,nb", [(2, (1300, 300, 450, 11), 512), (3, (4480, 920, 1344, 12), 1024)])
def test_wide_bulk_tiles_across_ranks(world, shape, nb):
    """The 128 x 256 bulk tiles (the default for launches over more than 20480 rows: sizes no test with several ranks on one
    GPU reaches) forced on at a small size across ranks: block-cyclic column tiles of 256, the direction against the oracle
    and bit for bit the one of the 128 x 128 tiles."""
    import torch.multiprocessing as mp
    n, me, mi, seed = shape
    res = {}
    # (one message per panel: with the two-message protocol a rank's share of a bulk update comes as the panel it factors next,
    #  alone, and the rest -- at these sizes every launch would be ONE panel, which takes the 32-row kernel of a head; reserve_cus = 0
    #  lifts the row threshold of the wide tiles)
    for name, opts in (("wide", {"bulk_bn": 256, "persist_rows": 0, "reserve_cus": 0, "dist_slices": 0}),
                       ("narrow", {"bulk_bn": 128, "dist_slices": 0})):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), shape, nb, "native", out, opts), nprocs=world, join=True)
        res[name] = {r: out[r] for r in range(world)}
    qp = make_qp(n, me, mi, seed)
    ref, _, _, _ = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                   qp["mu"], n, me, mi, regularise=False)
    for r in range(world):
        dz, st = res["wide"][r][0], res["wide"][r][1]
        assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= 1e-10
        assert st["n_neg"] == me + mi and st["n_zero"] == 0
        assert np.array_equal(dz, res["narrow"][r][0])
    assert sum(res["wide"][r][5]["instances"][256]["launches"] for r in range(world)) > 0       # (the wide instance did run)
    assert sum(res["narrow"][r][5]["instances"][256]["launches"] for r in range(world)) == 0


def test_native_and_python_drivers_agree_bitwise():
    """Same kernels in the same order on every column: the direction does not depend on who drives the schedule."""
    import torch.multiprocessing as mp
    shape, nb, world = (900, 200, 300, 8), 256, 2
    res = {}
    for mode in ("native", "python-lookahead"):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), shape, nb, mode, out), nprocs=world, join=True)
        res[mode] = out[0][0]
    assert np.array_equal(res["native"], res["python-lookahead"])


@pytest.mark.parametrize("shape", [(4096, 1024, 1536, 3), (3000, 700, 1100, 4), (2048, 0, 0, 5)])
def test_wide_panels_match_the_group_schedule_bitwise(shape):
    """Per-panel schedule at nb = 512 / 1024 (what several GPUs run): the owner's wide panel is factored as a block of
    256-column sub-panels and swept sub-panel by sub-panel -- the launches of the single-rank group schedule at nb = 256 on
    the same entries in the same order, so the direction has the same bits; with all stages of a wide panel in one launch
    (wide_sub = 0, the round-2 form) it agrees to rounding."""
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.dist import DistNewton
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)

    def run(nb, wide_sub, per_panel, **opts):
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        core.set_option("wide_sub", wide_sub)
        for k_, v_ in opts.items():
            core.set_option(k_, v_)
        core.set_option("sweep_persist", 0)          # (the per-panel sweeps on both sides: the one-launch backward sweep sums in another order)
        dz, st = (core.step_dist(0.0, 0.0) if per_panel else core.step(0.0, 0.0))
        dz = dz.clone()
        core.close()
        return dz, st

    ref, st0 = run(256, 256, False)
    for nb in (512, 1024):
        wide, st1 = run(nb, 256, True)
        one, st2 = run(nb, 0, True)
        assert st0["n_neg"] == st1["n_neg"] == st2["n_neg"] == me + mi
        assert torch.equal(wide, ref)
        assert float((one - ref).norm() / ref.norm()) <= 1e-12


def test_dist_driver_world1_matches_fused_step():
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.dist import DistNewton
    n, me, mi = 700, 200, 300
    qp = make_qp(n, me, mi, 8)
    core = NewtonCore(n, me, mi, device=0, nb=256)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz0, st0 = core.step(0.0, 0.0)
    for native in (True, False):
        dz1, st1 = DistNewton(core, native=native).step(0.0, 0.0)
        assert st0["n_neg"] == st1["n_neg"] == me + mi
        assert float((dz0 - dz1).norm() / dz0.norm()) <= 1e-12
    core.set_option("dist_selfmsg", 1)                   # one rank, but every panel is packed and "sent"
    core.set_option("profile", 1)
    dz2, st2 = core.step_dist(0.0, 0.0)
    tm = core.dist_timings()
    fact, _ = _factor_bytes(n, me, mi, 256)
    assert tm["bytes"] == fact and tm["messages"] > 0 and tm["pack_ms"] > 0.0 and tm["chain_ms"] > 0.0
    assert float((dz0 - dz2).norm() / dz0.norm()) <= 1e-12
    # the forward sweep that trails the factorisation (step_dist) runs the same kernels in the same order as the sweep of
    # solve_dist: the same bits, and the solve that follows is shorter by that sweep
    import torch
    core.set_option("fuse_forward", 0)
    dz3, _ = core.step_dist(0.0, 0.0)
    tm3 = core.dist_timings()
    core.set_option("fuse_forward", 1)
    dz4, _ = core.step_dist(0.0, 0.0)
    assert torch.equal(dz2, dz3) and torch.equal(dz3, dz4)
    assert core.dist_timings()["solve_ms"] < tm3["solve_ms"]


def _rccl_worker(rank, port, shape, nb, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import ctypes
        from pyipm_amd.newton import NewtonCore
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz0, st0 = core.step(0.0, 0.0)
        # handle-owned communicator: id from the library, communicator of one rank, every panel message
        # really goes through ncclBroadcast on the collective stream, the statistics through ncclAllReduce
        lib = core.lib
        lib.pyipm_newton_rccl_library(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode())
        idbuf = (ctypes.c_char * 128)()
        assert lib.pyipm_newton_comm_unique_id(ctypes.cast(idbuf, ctypes.c_void_p)) == 0
        core.comm_init(bytes(idbuf))
        core.set_option("dist_selfmsg", 1)
        core.set_option("profile", 1)
        dz1, st1 = core.step_dist(0.0, 0.0)
        dz2, st2 = core.step_dist(0.0, 0.0)             # buffers / streams / events reused on the second call
        tm = core.dist_timings()
        torch.cuda.synchronize()
        out[0] = (float((dz0 - dz1).norm() / dz0.norm()), float((dz1 - dz2).norm()), st1["n_neg"], tm, dist.get_backend())
    finally:
        dist.destroy_process_group()


def test_handle_owned_rccl_communicator_single_rank():
    """The multi-GPU bench path -- RCCL communicator owned by the handle, asynchronous per-panel broadcasts on the
    collective stream, owner factoring on the side stream -- run for real on the one GPU this box has."""
    import torch.multiprocessing as mp
    shape, nb = (900, 200, 300, 8), 256
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(_free_port(), shape, nb, out), nprocs=1, join=True)
    err, rep, n_neg, tm, backend = out[0]
    assert backend == "nccl"
    assert err <= 1e-12 and rep == 0.0 and n_neg == shape[1] + shape[2]
    assert tm["bytes"] == _factor_bytes(shape[0], shape[1], shape[2], nb)[0] and tm["bcast_ms"] > 0.0


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per process), with
    both ranks on the one GPU over gloo (PYIPM_BENCH_SHARE_GPU=1): row-sharded staging, the library's distributed
    driver, the JSON line with n_gpus = 2 and the per-phase timings."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYIPM_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nvar", "1536", "--neq", "256", "--nineq", "640", "--nb", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["kkt_dim"] == 1536 + 2 * 640 + 256
    assert d["backward_error"] <= 1e-12 and d["inertia"]["n_neg"] == 256 + 640 and d["inertia"]["n_zero"] == 0
    ph = d["dist_phases_per_step"]
    assert ph["messages"] > 0 and ph["bytes"] > 0 and ph["factor_ms"] > 0 and ph["chain_ms"] > 0


def _rccl_world_worker(rank, world, port, shape, nb, out):
    """One rank per GPU, nccl process group, handle-owned RCCL communicator (what bench.py --gpus N runs)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from pyipm_amd.newton import NewtonCore
        from pyipm_amd.dist import DistNewton
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        core = NewtonCore(n, me, mi, device=rank, nb=nb, world=world, rank=rank)
        rows = core.owned_rows()
        core.stage_blocks_owned(qp["d2L"][rows], qp["Je"][rows] if me else None, qp["Ji"][rows] if mi else None)
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        core.set_option("profile", 1)
        core.set_option("dist_sag_min_bytes", 1024)      # every panel message in the scatter + all-gather form (three ranks or more)
        drv = DistNewton(core, native=True)
        dz, st = drv.step(0.0, 0.0)
        dz2, _ = drv.step(0.0, 0.0)
        tm = core.dist_timings()
        mode = core.comm_bcast_mode()
        core.set_option("dist_sag", 0)                    # ... and once more through ncclBroadcast: the same bits
        dz3, _ = drv.step(0.0, 0.0)
        torch.cuda.synchronize()
        out[rank] = (dz.cpu().numpy(), float((dz - dz2).norm()) + float((dz - dz3).norm()), st, core.comm_ranks(), tm, mode)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_world_matches_single_rank(world):
    """Runs wherever the box has at least `world` GPUs (skips on the one-GPU development box): handle-owned RCCL
    communicator over `world` real devices, direction equal to the one-rank direction to 1e-12, RCCL itself
    counting `world` ranks, bytes on the wire as predicted; with three ranks or more every panel message travels as
    scatter + all-gather (forced on for these small panels) and once more through ncclBroadcast: the same bits."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, box has %d" % (world, torch.cuda.device_count()))
    from pyipm_amd.newton import NewtonCore
    shape, nb = (1500, 300, 500, 8), 256
    n, me, mi, seed = shape
    qp = make_qp(n, me, mi, seed)
    core = NewtonCore(n, me, mi, device=0, nb=nb)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz0, st0 = core.step(0.0, 0.0)
    dz0 = dz0.cpu().numpy()
    del core
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_world_worker, args=(world, _free_port(), shape, nb, out), nprocs=world, join=True)
    for r in range(world):
        dz, rep, st, ranks, tm, mode = out[r]
        assert ranks == world
        assert mode == (1 if world >= 3 else 0)              # scatter + all-gather passed comm_init's self-test (three ranks or more)
        assert np.linalg.norm(dz - dz0) <= 1e-12 * np.linalg.norm(dz0)
        assert rep == 0.0 and st["n_neg"] == me + mi and st["n_zero"] == 0
        assert tm["bytes"] == _factor_bytes(n, me, mi, nb)[0]      # every rank takes part in every broadcast


def test_bench_self_launches_for_several_gpus():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks (it used to exit).  On a
    one-GPU box both ranks share the device over gloo; on a multi-GPU box this is the real RCCL path and the line
    says how many ranks RCCL counted."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    multi = torch.cuda.device_count() >= 2
    if not multi:
        env["PYIPM_BENCH_SHARE_GPU"] = "1"
    # (nb = 1024, the multi-GPU default from KKT dimension 65536 on -- five panels here, the last one 640 wide, each factored as a
    # block of sub-panels; below that size bench.py takes nb = 256: tools/rank_replay.py)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--nvar", "2560", "--neq", "384", "--nineq", "896", "--nb", "1024", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backward_error"] <= 1e-12 and d["config"]["nb"] == 1024
    assert d["inertia"]["n_neg"] == 384 + 896 and d["inertia"]["n_zero"] == 0
    assert d["rccl_ranks"] == (2 if multi else 0)


def _cond_worker(rank, world, port, shape, nb, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyipm_amd.newton import NewtonCore
        from pyipm_amd.dist import DistNewton
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        core = NewtonCore(n, me, mi, device=0, nb=nb, world=world, rank=rank)
        core.set_option("condensed", 1)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])                 # full blocks: a column of Ji Sigma Ji' takes every row of Ji
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        drv = DistNewton(core, native=True)
        g = drv.residual()
        dz, st = drv.step(0.0, 0.0, refine=1)
        raw = dz.clone()
        raw[n + mi:] *= -1.0
        berr = float((core.matvec_dist(raw) - g).norm() / g.norm())
        dz2 = core.solve_dist(flip=True, refine=-1)
        torch.cuda.synchronize()
        out[rank] = (dz.cpu().numpy(), st, core.ncols_local, berr, core.solve_info(), dz2.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,nb", [(2, (900, 200, 300, 8), 256), (3, (700, 150, 260, 9), 128), (2, (1400, 0, 400, 10), 128)])
def test_condensed_option_across_ranks(world, shape, nb):
    """SURVEY 8f rank 2 x 8e (missing 3): the condensed KKT system [[H + Ji Sigma Ji', Je], [Je', 0]] factored by the
    per-panel schedule over several ranks (sharing the one GPU here, exchange over gloo callbacks): every rank assembles the
    columns it owns -- the Gram launch included --, the sweeps run on the condensed geometry between a replicated reduce and
    expand, refinement against the full blocks across the ranks.  Direction of the FULL system <= 1e-10 of the oracle's,
    full-system inertia, the one-rank condensed direction to rounding."""
    import torch.multiprocessing as mp
    from pyipm_amd.newton import NewtonCore
    n, me, mi, seed = shape
    N = n + 2 * mi + me
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cond_worker, args=(world, _free_port(), shape, nb, out), nprocs=world, join=True)
    qp = make_qp(n, me, mi, seed)
    ref, _, _, _ = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                   qp["mu"], n, me, mi, regularise=False)
    one = NewtonCore(n, me, mi, device=0, nb=nb)
    one.set_option("condensed", 1)
    one.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    one.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz1, st1 = one.step(0.0, 0.0, refine=1)
    dz1 = dz1.cpu().numpy()
    cols = 0
    out = {r: out[r] for r in range(world)}
    for r in range(world):
        dz, st, ncl, berr, info, dz2 = out[r]
        assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= 1e-10
        assert np.linalg.norm(dz - dz1) / np.linalg.norm(dz1) <= 1e-12
        assert (st["n_neg"], st["n_pos"], st["n_zero"]) == (me + mi, N - me - mi, 0) == (st1["n_neg"], st1["n_pos"], st1["n_zero"])
        assert berr <= 1e-12 and info["converged"] and info["backward_error"] <= 1e-13
        assert np.linalg.norm(dz2 - ref) / np.linalg.norm(ref) <= 1e-10
        cols += ncl
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0])
    one.close()                                       # (a handle left to the garbage collector would be destroyed a second time
                                                      #  in the fork that the next test's mp.Manager() makes of this process)


def test_condensed_across_ranks_needs_full_blocks():
    """Row-sharded staging cannot feed the Gram launch (a column of Ji Sigma Ji' takes every row of Ji): a clear error."""
    from pyipm_amd.newton import NewtonCore, NewtonError
    n, me, mi = 600, 100, 200
    qp = make_qp(n, me, mi, 3)
    c2 = NewtonCore(n, me, mi, device=0, nb=128, world=2, rank=0)
    try:
        c2.set_option("condensed", 1)
        rows = c2.owned_rows()
        c2.stage_blocks_owned(qp["d2L"][rows], qp["Je"][rows], qp["Ji"][rows])
        c2.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        with pytest.raises(NewtonError, match="full blocks"):
            c2.assemble(0.0, 0.0)
    finally:
        c2.close()


# ---- round 5: the exchange code that had never run and the two-message protocol (item 1a) ---------------
def _run_world(world, shape, nb, opts=None, drv_kw=None, env=None, mode="native-sharded"):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, nb, mode, out, opts, drv_kw, env), nprocs=world, join=True)
    return {r: out[r] for r in range(world)}


def _n_slices(n, me, mi, nb, world):
    """Slices the geometry calls for (the library's sl(k) rule): panel k has a message, panel k + 1 exists, has another owner
    and is not inside the slack block, panel k is not inside it either; slice 2 only when panel k + 2 exists."""
    N = n + 2 * mi + me
    Npad = ((N + 127) // 128) * 128
    npan = (Npad + nb - 1) // nb
    in_s = lambda p: mi > 0 and p * nb >= n and min((p + 1) * nb, Npad) <= n + mi          # noqa: E731
    count = 0
    for k in range(npan - 1):
        w = min(nb, Npad - k * nb)
        if Npad - (k * nb + w) <= 0 or in_s(k) or in_s(k + 1) or world < 2:
            continue
        count += 1 + (1 if k + 2 < npan else 0)
    return count


@pytest.mark.parametrize("world,shape,nb", [(2, (900, 200, 300, 8), 256), (3, (700, 150, 260, 9), 128), (4, (1500, 300, 500, 10), 256),
                                            (3, (2000, 400, 600, 12), 1024)])
def test_exchange_forms_give_the_same_bits(world, shape, nb):
    """Every wire form of the distributed factorisation on 2 - 4 ranks sharing the GPU, exchange over gloo callbacks:
      A  slices point to point + panel messages as scatter + all-gather (every size) + every operation serialised through the
         collective stream (what the RCCL path does on several GPUs: sag_bcast's slicing with a count that does not divide by
         the number of ranks, the stream hop, the self-test that switches the form on);
      B  slices as broadcasts (an exchange without point-to-point callbacks), plain panel broadcasts;
      C  one message per panel (dist_slices = 0), scatter + all-gather;
      D  one message per panel, plain broadcast -- the protocol of rounds 1 - 4;
      E  as B with dist_slices = 1: the rows of the second slice as a k_panel_rest launch behind the chain (round 5) instead of
         units of the chain's own launch (round 6, the default in A and B).
    The same direction bit for bit from all four, on every rank; message and byte counts per form as predicted."""
    n, me, mi, seed = shape
    p2p = {"p2p": True, "serialize": True, "selftest": True}
    nop2p = {"p2p": False}
    runs = {"A": _run_world(world, shape, nb, {"dist_sag_min_bytes": 8}, p2p),
            "B": _run_world(world, shape, nb, None, nop2p),
            "C": _run_world(world, shape, nb, {"dist_slices": 0, "dist_sag_min_bytes": 8}, {"p2p": True, "serialize": False, "selftest": True}),
            "D": _run_world(world, shape, nb, {"dist_slices": 0}, nop2p),
            "E": _run_world(world, shape, nb, {"dist_slices": 1}, nop2p)}        # (B without the second slice's rows in the chain's launch)
    qp = make_qp(n, me, mi, seed)
    ref, _, _, _ = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                   qp["mu"], n, me, mi, regularise=False)
    base = runs["D"][0][0]
    assert np.linalg.norm(base - ref) / np.linalg.norm(ref) <= 1e-10
    fact, _ = _factor_bytes(n, me, mi, nb)
    nsl = _n_slices(n, me, mi, nb, world)
    for name, res in runs.items():
        for r in range(world):
            dz, st, _, _, _, extra = res[r]
            assert np.array_equal(dz, base), (name, r)
            assert st["n_neg"] == me + mi and st["n_zero"] == 0
            assert extra["timings"]["bytes"] == fact                         # the panel messages themselves never change
            w = extra["wire"]
            sag = name in ("A", "C") and world >= 3
            assert extra["bcast_mode"] == (1 if sag else 0), (name, extra["bcast_mode"])
            nmsg = extra["timings"]["messages"]
            if sag:
                assert (w["sag_messages"], w["sag_bytes"], w["bcast_messages"]) == (nmsg, fact, 0), (name, w)
                assert w["allgathers"] == nmsg
            else:
                assert (w["bcast_messages"], w["bcast_bytes"], w["sag_messages"], w["allgathers"]) == (nmsg, fact, 0, 0), (name, w)
            if name in ("C", "D"):
                assert w["slice_messages"] == 0 and w["slices_as_broadcast"] == 0
            if name == "A":
                assert w["stream_hops"] > 0                                   # the sweeps' exchanges hop through the collective stream
            if name in ("B", "D", "E"):
                assert w["stream_hops"] == 0
        # the second slice's rows rode in the chain's launch where the slices arrive on the collective stream (A, B), nowhere else
        xr = sum(res[r][5]["wire"]["chains_with_extra_rows"] for r in range(world))
        assert (xr > 0) if name in ("A", "B") else (xr == 0), (name, xr)
        if name == "A":
            assert sum(res[r][5]["wire"]["slice_messages"] for r in range(world)) == 2 * nsl     # counted by sender and receiver
            assert all(res[r][5]["wire"]["slices_as_broadcast"] == 0 for r in range(world))
        if name == "B":
            assert all(res[r][5]["wire"]["slices_as_broadcast"] == nsl for r in range(world))
    assert nsl > 0


def test_selftest_agreement_one_rank_opts_out():
    """PYIPM_DIST_SAG=0 on ONE rank of three: the ranks agree on NOT using the scatter + all-gather form (a rank entering the
    self-test alone would wait for ever), every panel message travels as a broadcast, the result does not change."""
    shape, nb, world = (700, 150, 260, 9), 128, 3
    res = _run_world(world, shape, nb, {"dist_sag_min_bytes": 8}, {"p2p": True, "serialize": True, "selftest": True},
                     env={1: {"PYIPM_DIST_SAG": "0"}})
    ref = _run_world(world, shape, nb, {"dist_slices": 0}, {"p2p": False})
    for r in range(world):
        assert res[r][5]["bcast_mode"] == 0 and res[r][5]["wire"]["sag_messages"] == 0
        assert np.array_equal(res[r][0], ref[0][0])


def test_a_stalled_panel_message_is_an_error_code_not_a_hang():
    """VERDICT r5 item 2b.  RCCL has no per-operation timeout: a collective a peer never joins leaves its stream stopped and the
    host inside a synchronisation for ever.  The distributed step now waits for the device with a bound (set_option
    ("dist_timeout_s")) and reads which panel's message / bulk update / chain the device did complete from progress words a
    one-thread kernel writes behind each of them (csrc/dist_impl.hpp:bounded_wait).  The test hook debug_fault = 3 puts a
    kernel that spins for three time bounds in front of the middle panel's message: the step returns PYIPM_E_COMM naming that
    panel, the handle refuses further steps, and closing it does not hang.  (pyipm.py:1720-1721 has no counterpart: one process.)"""
    import time
    from pyipm_amd.newton import NewtonCore, NewtonError
    n, me, mi = 1500, 200, 300
    qp = make_qp(n, me, mi, 3)
    core = NewtonCore(n, me, mi, device=0, nb=256)
    core.set_option("expert", 1)
    core.set_option("dist_selfmsg", 1)                      # one rank: pack and "send" every panel anyway
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz0, st0 = core.step_dist(0.0, 0.0)                     # a healthy step first
    assert st0["n_neg"] == me + mi
    core.set_option("dist_timeout_s", 0.5)
    core.set_option("debug_fault", 3)
    t0 = time.perf_counter()
    with pytest.raises(NewtonError) as ei:
        core.step_dist(0.0, 0.0)
    waited = time.perf_counter() - t0
    assert ei.value.code == -6, ei.value                    # PYIPM_E_COMM
    npanels = core.npanels                                  # (9 panels; the middle one, panel 4, lies in the x block: it has a message)
    assert "panel messages before panel %d," % (npanels // 2) in str(ei.value), str(ei.value)
    assert 0.4 <= waited <= 1.4, waited                     # the bound, not the stall (1.5 s)
    with pytest.raises(NewtonError):
        core.step_dist(0.0, 0.0)                            # the handle knows its streams are not to be trusted
    core.close()                                            # ... and goes away without hanging (the stall ends by itself here)
    # a fresh handle on the same problem gives the healthy step's bits
    core = NewtonCore(n, me, mi, device=0, nb=256)
    core.set_option("dist_selfmsg", 1)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz1, _ = core.step_dist(0.0, 0.0)
    import torch
    assert torch.equal(dz0, dz1)
    core.close()


def test_bench_ladder_keeps_a_number_when_a_faster_wire_form_stalls():
    """VERDICT r5 item 2a.  Across GPUs bench.py times every wire form of the distributed factorisation, safest first (plain
    broadcast -> scatter + all-gather -> slices on one communicator -> the same with the second slice's rows inside the chain's
    launch -> slices on a second communicator), and from the second
    rung on its watchdog holds the best completed result.  Two ranks on the one GPU over gloo; the third rung is made to stall
    (PYIPM_BENCH_LADDER_STALL=2: debug_fault = 3 with the step's own time bound off, i.e. a message that does not complete for
    30 s against a 8 s watch): the run ends with status 0 and ONE line whose value is the best of the two rungs that completed,
    naming the rung that stalled.  Without a stall the line lists all five rungs and names the one the timed region ran on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--nvar", "1536", "--neq", "256", "--nineq", "640", "--nb", "256", "--no-cpu-baseline"]
    env = dict(os.environ, PYIPM_BENCH_SHARE_GPU="1")
    out = subprocess.run(base + ["--master-port", str(_free_port())] + tail, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert len(d["ladder"]) == 5 and d["wire_form"] in [r["wire_form"] for r in d["ladder"]] and d["value"] > 0
    assert all(r["value"] > 0 for r in d["ladder"])
    env = dict(env, PYIPM_BENCH_LADDER_STALL="2", PYIPM_BENCH_LADDER_WATCH="8")
    out = subprocess.run(base + ["--master-port", str(_free_port())] + tail, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] is not None and d["value"] > 0 and "error" not in d
    assert "rung 2" in d["ladder_stalled_at"] and len(d["ladder"]) == 2
    assert d["value"] == max(r["value"] for r in d["ladder"]) and d["wire_form"] in [r["wire_form"] for r in d["ladder"]]
    assert d["n_gpus"] == 2 and d["metric"] == "newton_steps_per_sec" and d["config"]["kkt_dim"] == 1536 + 2 * 640 + 256
