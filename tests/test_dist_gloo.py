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


# ---------------------------------------------------------------------- row-sharded L-BFGS direction
def _lbfgs_case(n, me, mi, m, seed):
    from oracle import lbfgs_oracle as lo
    rng = np.random.default_rng(seed)
    qp = make_qp(n, me, mi, seed)
    eps = float(np.finfo(float).eps)
    zeta, S, Y, SS, L, D, fail = lo.lbfgs_init(n)
    x_old = rng.standard_normal(n)
    Mq = rng.standard_normal((n, 8)) / 3.0
    for _ in range(m):
        x_new = x_old + rng.standard_normal(n) / np.sqrt(n)
        hv = lambda v: Mq @ (Mq.T @ v) + 0.5 * v           # noqa: E731
        zeta, S, Y, SS, L, D, fail = lo.lbfgs_update(x_old, x_new, -hv(x_old), -hv(x_new), zeta, S, Y, SS, L, D, fail,
                                                     n, bool(me or mi), max(m, 1), eps)
        x_old = x_new
    g = rng.standard_normal(n + 2 * mi + me)
    return qp, zeta, S, Y, SS, L, D, g


def _lbfgs_worker(rank, world, port, shape, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from backends import lbfgs_direction_row_shard
        n, me, mi, m, seed = shape
        qp, zeta, S, Y, SS, L, D, g = _lbfgs_case(n, me, mi, m, seed)
        cut = [(n * r) // world for r in range(world + 1)]
        a, b = cut[rank], cut[rank + 1]
        nbytes = [0]

        def allreduce(arr):
            t = torch.from_numpy(arr)                      # shares memory: in place
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            nbytes[0] += arr.size * 8

        dz = lbfgs_direction_row_shard(allreduce, qp["Je"][a:b] if me else None, qp["Ji"][a:b] if mi else None,
                                       np.concatenate([g[a:b], g[n:]]), qp["s"] if mi else np.zeros(0),
                                       qp["lam"] if (me or mi) else np.zeros(0), zeta, S[a:b], Y[a:b], SS, L, D, me, mi)
        out[rank] = (a, b, dz, nbytes[0])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (300, 30, 70, 5, 1)), (3, (250, 0, 60, 4, 2)), (2, (200, 40, 0, 3, 3)),
                                         (4, (320, 0, 0, 6, 4)), (2, (150, 20, 30, 0, 5))])
def test_lbfgs_row_sharding_gloo(world, shape):
    """The sharding scheme of the L-BFGS direction (rows of J, three all-reduces: include/pyipm_lbfgs.h) on CPU: a
    NumPy model of one rank (tests/backends.py) with the sums at the places the HIP library calls its callback,
    against the oracle's unsharded direction.  The HIP ranks themselves are covered by tests/test_gpu_lbfgs.py."""
    from oracle import lbfgs_oracle as lo
    n, me, mi, m, seed = shape
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_lbfgs_worker, args=(world, _free_port(), shape, out), nprocs=world, join=True)
    qp, zeta, S, Y, SS, L, D, g = _lbfgs_case(n, me, mi, m, seed)
    ref = lo.direction(g, zeta, S, Y, SS, L, D, Je=qp["Je"] if me else None, Ji=qp["Ji"] if mi else None,
                       s=qp["s"] if mi else np.zeros(0), lda=qp["lam"] if (me or mi) else np.zeros(0), reg=0.0)
    scale = np.linalg.norm(ref)
    p, r, rr = me + mi, 2 * m, 2 * m + 1
    for rk in range(world):
        a, b, dz, nbytes = out[rk]
        assert np.linalg.norm(dz[:b - a] - ref[a:b]) <= 1e-10 * scale
        assert np.linalg.norm(dz[b - a:] - ref[n:]) <= 1e-10 * scale
        assert nbytes == 8 * ((p * p + p * rr + (r * rr if m else 0)) if p else r)
