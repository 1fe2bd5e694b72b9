"""Where does the FIRST Newton direction of a solve spend its extra time (one-time schedule construction, streams, events,
device allocations)?  Times assemble / factor / solve / rcond of calls 1..4 on a fresh handle at the headline size.
Usage: python tools/first_call.py [--nvar N --neq ME --nineq MI]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvar", type=int, default=16384)
    ap.add_argument("--neq", type=int, default=4096)
    ap.add_argument("--nineq", type=int, default=6144)
    args = ap.parse_args()
    import torch
    from bench import make_qp_device
    from pyipm_amd.newton import NewtonCore
    dev = torch.device("cuda", 0)
    n, me, mi = args.nvar, args.neq, args.nineq
    qp = make_qp_device(n, me, mi, 0, dev)
    torch.cuda.synchronize()

    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, r

    t_create, core = timed(lambda: NewtonCore(n, me, mi, device=0))
    t_stage, _ = timed(lambda: (core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"]),
                                core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])))
    rows = []
    for call in range(4):
        r = {"call": call + 1}
        r["residual_ms"], _ = timed(core.residual)
        r["assemble_ms"], _ = timed(lambda: core.assemble(0.0, 0.0))
        r["factor_ms"], _ = timed(core.factor)
        r["rcond_adaptive_ms"], est = timed(lambda: core.rcond(-1, -1))
        r["rcond_adaptive"] = est["rcond"]
        r["rcond_fixed_ms"], est = timed(lambda: core.rcond())
        r["rcond_fixed"] = est["rcond"]
        r["solve_ms"], _ = timed(lambda: core.solve(flip=True))
        r["step_lengths_ms"], _ = timed(lambda: core.step_lengths(0.995))
        r["merit_info_ms"], _ = timed(core.merit_info)
        r["merit_ray64_ms"], _ = timed(lambda: core.merit_ray([0.9 * 0.995 ** k for k in range(64)], 10.0, 0.2))
        r["merit_ray64_again_ms"], _ = timed(lambda: core.merit_ray([0.9 * 0.995 ** k for k in range(64)], 10.0, 0.2))
        rows.append(r)
    print(json.dumps({"create_ms": t_create, "stage_ms": t_stage, "calls": rows}))


if __name__ == "__main__":
    main()
