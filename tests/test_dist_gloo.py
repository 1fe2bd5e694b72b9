"""world_size-2 (and 3) gloo tests of the multi-GPU host orchestration (pyipm_amd/dist.py) on CPU.
The per-panel HIP operations are stood in by tests/backends.ModelCore (NumPy model of one rank);
what is under test is the product's DistNewton: ownership, message sequencing, stats reduction,
owner-by-owner substitutions.  Checked against the oracle's LU solve."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import newton_oracle as orc
from pyipm_amd.problems import make_qp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shape, nb, lookahead, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from backends import ModelCore
        from pyipm_amd.dist import DistNewton
        n, me, mi, seed = shape
        qp = make_qp(n, me, mi, seed)
        H = orc.kkt_matrix(qp["d2L"], qp["Je"], qp["Ji"], qp["s"], qp["lam"], n, me, mi)
        g = -orc.kkt_residual(qp["df"], qp["Je"], qp["Ji"], qp["ce"], qp["ci"], qp["s"], qp["lam"], qp["mu"], n, me, mi)
        core = ModelCore(n, me, mi, H, g, nb=nb, world=world, rank=rank)
        drv = DistNewton(core)
        drv.lookahead = lookahead
        dz, st = drv.step(0.0, 0.0)
        ref = orc.flip_multipliers(orc.sym_solve(H, g.reshape(-1, 1)).reshape(-1), n, mi)
        err = float(np.linalg.norm(dz.numpy() - ref) / np.linalg.norm(ref))
        out[rank] = (err, st["n_neg"], st["n_zero"], st["n_pos"], drv.bytes_broadcast, core.ncols_local)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lookahead", [False, True])
@pytest.mark.parametrize("world,shape,nb", [(2, (150, 40, 60, 3), 128), (2, (300, 90, 120, 4), 128),
                                            (3, (260, 50, 100, 5), 128), (2, (200, 0, 150, 6), 256),
                                            (4, (420, 100, 180, 7), 128), (8, (640, 128, 200, 8), 128)])
def test_dist_newton_gloo(world, shape, nb, lookahead):
    n, me, mi, _ = shape
    N = n + 2 * mi + me
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), shape, nb, lookahead, out), nprocs=world, join=True)
    assert len(out) == world
    cols = 0
    for r in range(world):
        err, neg, zero, pos, nbytes, ncl = out[r]
        assert err <= 1e-10, (r, err)
        assert neg == me + mi and zero == 0 and pos == N - me - mi      # stats are all-reduced: same everywhere
        cols += ncl
    assert cols == ((N + 127) // 128) * 128                               # every column owned exactly once
    assert out[0][4] == out[1][4] > 0                                     # same bytes on the wire on every rank
