"""GPU tests of the distributed per-panel HIP path.  The GPU box has ONE device, so world_size-2/3
runs put every rank on cuda:0 and ride a gloo group with host staging (RCCL refuses duplicate
devices); the HIP kernels, block-cyclic column maps, pack/unpack messages and the non-owner
L-rebuild are exactly what the multi-GPU bench uses."""
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


def _worker(rank, world, port, shape, nb, lookahead, out):
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
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        drv = DistNewton(core)
        drv.lookahead = lookahead
        dz, st = drv.step(0.0, 0.0)
        torch.cuda.synchronize()
        out[rank] = (dz.cpu().numpy(), st, core.ncols_local, drv.bytes_broadcast)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lookahead", [False, True])
@pytest.mark.parametrize("world,shape,nb", [(2, (300, 100, 150, 7), 128), (2, (900, 200, 300, 8), 256),
                                            (3, (700, 150, 260, 9), 128), (2, (1400, 0, 400, 10), 128)])
def test_two_ranks_one_gpu(world, shape, nb, lookahead):
    import torch.multiprocessing as mp
    n, me, mi, seed = shape
    N = n + 2 * mi + me
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, nb, lookahead, out), nprocs=world, join=True)
    qp = make_qp(n, me, mi, seed)
    ref, _, _, _ = orc.newton_step(qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"],
                                   qp["mu"], n, me, mi, regularise=False)
    cols = 0
    for r in range(world):
        dz, st, ncl, _ = out[r]
        assert np.linalg.norm(dz - ref) / np.linalg.norm(ref) <= 1e-10
        assert st["n_neg"] == me + mi and st["n_zero"] == 0 and st["n_pos"] == N - me - mi
        cols += ncl
    assert cols == ((N + 127) // 128) * 128
    assert np.array_equal(out[0][0], out[1][0])          # every rank ends with the same direction, bit for bit
    # wire traffic of the factorisation: one message per panel that has rows below it, EXCEPT panels inside the slack
    # block (every rank derives their contribution from s / lambda locally); the substitutions add 8 bytes per entry
    Npad = ((N + 127) // 128) * 128
    fact = 0
    for p in range(Npad // nb + (1 if Npad % nb else 0)):
        c0 = p * nb
        w = min(nb, Npad - c0)
        m = Npad - (c0 + w)
        in_s = c0 >= n and c0 + w <= n + mi
        if c0 + w <= n and mi:                               # panel inside the x block: whole 128-row tiles of slack rows stay home
            m -= max(0, (n + mi) // 128 * 128 - (n + 127) // 128 * 128)
        if m > 0 and not in_s:
            fact += 8 * (m * w + 2 * (w // 64) * 4096 + w // 64)
    solve = 8 * sum((Npad - p * nb) + min(nb, Npad - p * nb) for p in range((Npad + nb - 1) // nb))
    assert out[0][3] == fact + solve


def test_dist_driver_world1_matches_fused_step():
    import torch
    from pyipm_amd.newton import NewtonCore
    from pyipm_amd.dist import DistNewton
    n, me, mi = 700, 200, 300
    qp = make_qp(n, me, mi, 8)
    core = NewtonCore(n, me, mi, device=0, nb=256)
    core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    dz0, st0 = core.step(0.0, 0.0)
    dz1, st1 = DistNewton(core).step(0.0, 0.0)
    assert st0["n_neg"] == st1["n_neg"] == me + mi
    assert float((dz0 - dz1).norm() / dz0.norm()) <= 1e-12


def _nccl_worker(rank, port, shape, nb, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from pyipm_amd.newton import NewtonCore
        from pyipm_amd.dist import DistNewton
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        core = NewtonCore(n, me, mi, device=0, nb=nb)
        core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        dz0, st0 = core.step(0.0, 0.0)
        drv = DistNewton(core)
        drv.force_lookahead = True          # overlapped schedule: side stream + asynchronous RCCL broadcasts
        dz1, st1 = drv.step(0.0, 0.0)
        dz2, st2 = drv.step(0.0, 0.0)       # buffers / streams reused on the second call
        torch.cuda.synchronize()
        out[0] = (float((dz0 - dz1).norm() / dz0.norm()), float((dz1 - dz2).norm()), st1["n_neg"], drv.bytes_broadcast,
                  dist.get_backend())
    finally:
        dist.destroy_process_group()


def test_overlapped_schedule_on_rccl_single_rank():
    """The multi-GPU bench path (RCCL process group, asynchronous per-panel broadcast, owner factoring on a
    side stream) run for real on the one GPU this box has: a world of one rank owns every panel."""
    import torch.multiprocessing as mp
    shape, nb = (900, 200, 300, 8), 256
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nccl_worker, args=(_free_port(), shape, nb, out), nprocs=1, join=True)
    err, rep, n_neg, nbytes, backend = out[0]
    assert backend == "nccl"
    assert err <= 1e-12 and rep == 0.0 and n_neg == shape[1] + shape[2] and nbytes > 0
