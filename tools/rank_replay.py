"""Rank replay: what ONE rank of a W-rank factorisation does, measured on the one GPU this pool has.

NOT a scaling curve.  No run here ever had more than one device.  What is measured, and what is modelled:

  1. RECORD.  One full factorisation of the system with the per-panel schedule (the multi-GPU schedule, nb = 1024) on one
     rank; every panel is packed exactly as its owner would pack it for the wire and the message is kept (4.3 GB at
     N = 32768, 68.7 GB at N = 131072 -- parked in host memory there).
  2. REPLAY as rank r of W, for every r: the library's own distributed driver (pyipm_newton_step_dist, csrc/dist_impl.hpp)
     runs with world = W, rank = r on a handle that holds only that rank's columns (row-sharded staging).  Panels the rank
     OWNS are factored, packed and applied for real; FOREIGN panels "arrive" through the exchange callbacks from the
     recording, after a stated link model (below); the rank's own columns are updated for real, the sweeps run for real
     (their nb-long exchanges pay the model's latency).  At N = 32768 the columns the rank owns are compared BIT FOR BIT
     with the same columns of the recorded factorisation.
  3. LINK MODEL (stated, not measured): per message  t = latency + bytes / bandwidth  with
       "bcast-1link": ncclBroadcast as a ring / tree -- one xGMI link's bandwidth whatever W;
       "sag":         scatter + all-gather over the W - 1 links of the mesh (dist_impl.hpp:sag_bcast):
                      2 latencies + 2 (bytes / W) / bandwidth.
     Defaults: 75 GB/s per link and direction, 12 us per hop.  Change them with --link-gbs / --latency-us.
  4. WHAT A REPLAY CANNOT SEE: a foreign panel is available as soon as the link model allows -- its owner's chain is not
     waited for.  So a rank's wall time is a LOWER bound of its real one, and the owners' chains are added back as a
     second bound: across the ranks the chain of panel p + 1 cannot start before panel p has arrived, so
       T_chain = sum over panels of [unpack(p) + chain(p+1) + pack(p+1) + link(p+1)]       (each term measured on its owner)
     is a critical path on its own.  implied_step_ms = max(max_r wall_r, T_chain); the real step lies between that and
     the sum of the two.

Usage: python tools/rank_replay.py [--nvar N --neq ME --nineq MI] [--nb 1024] [--worlds 2,4,8] [--host-record]
Writes one JSON document to stdout."""
import argparse
import ctypes
import json
import os
os.environ.setdefault("PYIPM_EXPERT", "1")     # tools use expert switches (include/pyipm_newton.h)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvar", type=int, default=16384)
    ap.add_argument("--neq", type=int, default=4096)
    ap.add_argument("--nineq", type=int, default=6144)
    ap.add_argument("--nb", type=int, default=1024)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--link-gbs", type=float, default=75.0)
    ap.add_argument("--latency-us", type=float, default=12.0)
    ap.add_argument("--models", default="bcast-1link,sag")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--host-record", action="store_true", help="park the recorded messages in host memory while the recording "
                                                              "factorisation holds the whole matrix (N = 131072)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--ranks", default="", help="comma-separated ranks to replay (default: all)")
    ap.add_argument("--serialize", type=int, default=0, help="1: every exchange operation through the collective stream (one communicator)")
    ap.add_argument("--slices", type=int, default=2, help="2: the two-message protocol with the second slice's rows inside the chain's launch (round 6, default); 1: the two-message protocol of round 5; 0: one message per panel (rounds 1-4)")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import numpy as np
    import torch
    from bench import make_qp_device
    from pyipm_amd.newton import ALLGATHER_FN, ALLREDUCE_FN, BCAST_FN, RECV_FN, SEND_FN, NewtonCore, _RawDeviceArray
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n, me, mi, nb = args.nvar, args.neq, args.nineq, args.nb
    N = n + 2 * mi + me
    f64 = torch.float64
    qp = make_qp_device(n, me, mi, 0, dev)

    # ---- calibrate torch.cuda._sleep (cycles per microsecond) -------------------------------------------------------
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000); torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(20000000); e1.record(); torch.cuda.synchronize()
    cyc_per_us = 20000000 / (e0.elapsed_time(e1) * 1e3)

    # ---- 1. record ---------------------------------------------------------------------------------------------------
    t_rec = time.perf_counter()
    core1 = NewtonCore(n, me, mi, device=0, nb=nb)
    for kv in args.opt:
        k, v = kv.split("="); core1.set_option(k, float(v))
    core1.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
    core1.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
    core1.residual(); core1.assemble(0.0, 0.0); core1.factor_begin()
    Npad, npanels = core1.Npad, core1.npanels
    rec, host_rec = {}, {}
    for p in range(npanels):
        core1.factor_panel(p)
        k = core1.panel_msg_numel(p)
        w = min(nb, Npad - p * nb)
        if k > 0 and Npad - (p * nb + w) > 0:
            buf = torch.empty(k, dtype=f64, device=dev)
            core1.panel_pack(p, buf)
            if args.host_record:
                torch.cuda.synchronize()
                host_rec[p] = buf.cpu()
                del buf
            else:
                rec[p] = buf
        core1.trailing_update(p)
    st1 = core1.factor_end()
    torch.cuda.synchronize()
    verify = not args.no_verify and not args.host_record
    full = core1.kkt_storage() if verify else None          # (ncols, Npad): the recorded factor, column by column
    if not verify:
        core1.close(); del core1
        torch.cuda.empty_cache()
    if args.host_record:
        for p in sorted(host_rec):
            rec[p] = host_rec[p].to(dev)
        host_rec.clear()
    order = sorted(rec)
    msg_bytes = {p: rec[p].numel() * 8 for p in order}
    t_rec = time.perf_counter() - t_rec

    # ---- slices of the two-message protocol, cut out of the recorded panel messages --------------------------------------------
    # (round 5) slice j of panel p = the W rows of panel p + j (+ the tile inverses, tiles and flags with slice 1), column-major
    # with the slice's rows as leading dimension: exactly what csrc/pyipm_newton.hip:pack_slice sends.
    def in_s(p):
        return mi > 0 and p * nb >= n and min((p + 1) * nb, Npad) <= n + mi

    def hole(p):                                              # rows a panel inside the x block leaves home (whole 256-row blocks)
        if not mi or min((p + 1) * nb, Npad) > n:
            return 0, 0
        a, b = (n + 255) // 256 * 256, (n + mi) // 256 * 256
        return (a, b) if b > a else (0, 0)

    def has_slices(k):                                        # the library's sl(k)
        return (k in rec) and k + 1 < npanels and not in_s(k) and not in_s(k + 1)

    def slice_of(p, j):
        if p + j >= npanels:
            return None
        w, c1 = min(nb, Npad - p * nb), p * nb + min(nb, Npad - p * nb)
        h0, h1 = hole(p)
        m = Npad - c1 - (h1 - h0)
        r0, E = (p + j) * nb, min(nb, Npad - (p + j) * nb)
        Wm = rec[p][: m * w].view(w, m)                       # [column][message row]
        out = torch.zeros((w, E), dtype=f64, device=dev)
        for r in range(r0, r0 + E, 128):                      # (128-row pieces: the hole is 256-aligned)
            if h1 > h0 and h0 <= r < h1:
                continue                                      # rows nobody writes or reads
            mr = r - c1 - ((h1 - h0) if (h1 > h0 and r >= h1) else 0)
            out[:, r - r0: r - r0 + 128] = Wm[:, mr: mr + 128]
        parts = [out.reshape(-1)]
        if j == 1:
            parts.append(rec[p][m * w:])
        return torch.cat(parts)

    slice_bytes = {}
    for k in order:
        if has_slices(k):
            for j in (1, 2):
                if k + j < npanels:
                    w, E = min(nb, Npad - k * nb), min(nb, Npad - (k + j) * nb)
                    slice_bytes[(k, j)] = 8 * (E * w + (2 * (w // 64) * 4096 + w // 64 if j == 1 else 0))

    def link_ms(nbytes, W, model):
        lat, bw = args.latency_us * 1e-3, args.link_gbs * 1e6          # ms, bytes per ms
        if model == "sag" and W >= 3 and nbytes >= (4 << 20):
            return 2 * lat + 2.0 * (nbytes / W) / bw
        return lat + nbytes / bw

    def view(ptr, count):
        return torch.as_tensor(_RawDeviceArray(ptr, count), device=dev)

    def ext_stream(stream):
        return torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev)

    out = {"what": "rank replay on ONE GPU (tools/rank_replay.py): measured per-rank work + a stated link model; NOT a scaling curve",
           "workload": "n=%d me=%d mi=%d -> KKT dim N=%d, nb=%d, seed 0" % (n, me, mi, N, nb),
           "link_model": {"per_link_GB_per_s": args.link_gbs, "latency_us": args.latency_us,
                          "bcast-1link": "t = latency + bytes / bandwidth (ncclBroadcast: one link's bandwidth)",
                          "sag": "t = 2 latency + 2 (bytes / W) / bandwidth (scatter + all-gather over W - 1 links; messages >= 4 MiB, W >= 3)"},
           "recording": {"seconds": t_rec, "messages": len(order), "bytes": int(sum(msg_bytes.values())),
                         "inertia_n_neg": st1["n_neg"], "expected": me + mi},
           "sleep_cycles_per_us": cyc_per_us, "replays": []}

    # one GPU: the single-rank step of the same box for reference (grouped schedule, nb = 256), when it fits beside the recording
    for W in [int(w) for w in args.worlds.split(",")]:
        for model in args.models.split(","):
            ranks = []
            for r in ([int(v) for v in args.ranks.split(",")] if args.ranks else range(W)):
                core = NewtonCore(n, me, mi, device=0, nb=nb, world=W, rank=r)
                for kv in args.opt:
                    k, v = kv.split("="); core.set_option(k, float(v))
                rows = torch.from_numpy(core.owned_rows()).to(dev)
                core.stage_blocks_owned(qp["d2L"].index_select(0, rows), qp["Je"].index_select(0, rows) if me else None,
                                        qp["Ji"].index_select(0, rows) if mi else None)
                core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
                core.set_option("profile", 1)
                state = {"k": 0, "err": None, "link_ms": 0.0, "rk": 0, "slice_link_ms": 0.0}

                def bcast(user, ptr, nbytes, root, stream, state=state, W=W, model=model, r=r):
                    try:
                        with torch.cuda.stream(ext_stream(stream)):
                            k = state["k"]
                            if k < len(order) and nbytes == msg_bytes[order[k]]:          # the next panel message
                                p = order[k]; state["k"] = k + 1
                                t = link_ms(nbytes, W, model); state["link_ms"] += t
                                torch.cuda._sleep(int(t * 1e3 * cyc_per_us))
                                if root != r:
                                    view(ptr, nbytes // 8).copy_(rec[p], non_blocking=True)
                            else:                                                         # an nb-long exchange of the sweeps
                                torch.cuda._sleep(int(link_ms(nbytes, W, "bcast-1link") * 1e3 * cyc_per_us))
                        return 0
                    except Exception as e:          # nothing may propagate through the C frames
                        state["err"] = e
                        return 1

                def allreduce(user, ptr, count, op, stream, state=state, W=W):
                    try:
                        with torch.cuda.stream(ext_stream(stream)):
                            torch.cuda._sleep(int(2 * link_ms(8 * count, W, "bcast-1link") * 1e3 * cyc_per_us))
                        return 0
                    except Exception as e:
                        state["err"] = e
                        return 1

                # point-to-point half: the slices this rank receives come from the recording after one link's time (a slice is
                # one point-to-point message: one link, whatever the panel form); what it sends just occupies the stream that long
                def recv(user, ptr, nbytes, peer, stream, state=state, W=W, r=r):
                    try:
                        with torch.cuda.stream(ext_stream(stream)):
                            q = state["recvq"]
                            if state["rk"] >= len(q) or slice_bytes[q[state["rk"]]] != nbytes:
                                raise RuntimeError("replay: unexpected point-to-point receive of %d bytes" % nbytes)
                            pj = q[state["rk"]]; state["rk"] += 1
                            t = link_ms(nbytes, W, "bcast-1link"); state["slice_link_ms"] += t
                            torch.cuda._sleep(int(t * 1e3 * cyc_per_us))
                            view(ptr, nbytes // 8).copy_(slice_of(*pj), non_blocking=True)
                        return 0
                    except Exception as e:
                        state["err"] = e
                        return 1

                def send(user, ptr, nbytes, peer, stream, state=state, W=W):
                    try:
                        with torch.cuda.stream(ext_stream(stream)):
                            torch.cuda._sleep(int(link_ms(nbytes, W, "bcast-1link") * 1e3 * cyc_per_us))
                        return 0
                    except Exception as e:
                        state["err"] = e
                        return 1

                def allgather(user, sptr, rptr, nbytes, stream, state=state):
                    state["err"] = RuntimeError("replay: the panel form is modelled inside the broadcast callback")
                    return 1

                # the slices this rank receives, in the library's order: slice 1 of panel 0 first; then per slot k slice 2 of
                # panel k (to owner(k + 1)) and slice 1 of panel k + 1 (to owner(k + 2))
                recvq = [(0, 1)] if (args.slices and has_slices(0) and 1 % W == r) else []
                for k in range(npanels):
                    if args.slices and has_slices(k) and (k + 1) % W == r and (k, 2) in slice_bytes:
                        recvq.append((k, 2))
                    if args.slices and has_slices(k + 1) and (k + 2) % W == r:
                        recvq.append((k + 1, 1))
                state["recvq"] = recvq
                cb = (BCAST_FN(bcast), ALLREDUCE_FN(allreduce))
                core.set_exchange(*cb)
                cb2 = (SEND_FN(send), RECV_FN(recv), ALLGATHER_FN(allgather))
                core.set_exchange_p2p(*cb2, serialize=bool(args.serialize))     # (0: slices on their own stream -- the RCCL path's second communicator)
                core.set_option("dist_slices", int(args.slices))
                walls, tms, dts, wrs = [], [], [], []
                for it in range(args.steps + 1):
                    state["k"] = 0; state["link_ms"] = 0.0; state["rk"] = 0; state["slice_link_ms"] = 0.0
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    dz, st = core.step_dist(0.0, 0.0)
                    torch.cuda.synchronize()
                    if state["err"] is not None:
                        raise state["err"]
                    if it:                                             # (the first step builds the schedules)
                        walls.append((time.perf_counter() - t0) * 1e3); tms.append(core.timings()); dts.append(core.dist_timings())
                        lib_out = (ctypes.c_double * 12)(); core.lib.pyipm_newton_dist_wire(core.h, lib_out); wrs.append(list(lib_out))
                med = int(np.argsort(walls)[len(walls) // 2])
                row = {"rank": r, "wall_ms": walls[med], "factor_ms": dts[med]["factor_ms"], "chain_ms": dts[med]["chain_ms"],
                       "pack_ms": dts[med]["pack_ms"], "bcast_ms_incl_link_model": dts[med]["bcast_ms"],
                       "unpack_ms": dts[med]["unpack_ms"], "sweeps_ms": dts[med]["solve_ms"], "bulk_update_ms": tms[med]["trailing_ms"],
                       "bulk_update_tflops": (tms[med]["trailing_flops"] / 1e12) / max(tms[med]["trailing_ms"] * 1e-3, 1e-12),
                       "link_model_ms": state["link_ms"], "messages": dts[med]["messages"], "bytes": dts[med]["bytes"],
                       "owned_panels": len([p for p in range(npanels) if p % W == r]),
                       "rows_behind_the_chain_ms": wrs[med][10], "slice_messages": int(wrs[med][7]), "slice_bytes": int(wrs[med][8]),
                       "slice_link_model_ms": state["slice_link_ms"]}
                if verify:
                    local = core.kkt_storage()
                    same, lc = True, 0
                    for p in range(r, npanels, W):                   # (storage row = matrix column; entries on and below the diagonal)
                        w = min(nb, Npad - p * nb)
                        A_, B_ = local[lc:lc + w, p * nb:], full[p * nb:p * nb + w, p * nb:]
                        same = same and bool(torch.equal(torch.triu(A_[:, :w]), torch.triu(B_[:, :w]))) and bool(torch.equal(A_[:, w:], B_[:, w:]))
                        lc += w
                    row["owned_columns_bitwise_equal_to_recorded_factor"] = same
                ranks.append(row)
                core.close(); del core
                torch.cuda.empty_cache()
            # the owners' chain as a critical path of its own (docstring, 4.)
            n_msg = len(order)
            big_link = sum(link_ms(msg_bytes[p], W, model) for p in order)
            unpack_all = float(np.mean([x["unpack_ms"] / max(n_msg - x["owned_panels"], 1) for x in ranks])) * n_msg
            if args.slices:
                # two-message protocol: (1) the slice-1 chain -- per panel: unpack slice 1, head, tile chain, (slice 2:) unpack,
                # head, the rows of the next panel, pack slice 1 = the owners' `chain` spans -- plus one small message per panel;
                # (2) the panel messages reach the chain three panels later (panel k's message -> its receiver's rows of panel
                # k + 3 -> slice 2 of k + 1 -> slice 1 of k + 2 -> chain of k + 3): a third of [rows behind the chain + pack +
                # panel link + unpack + two slice links] per panel
                s1 = sum(link_ms(slice_bytes[(k, 1)], W, "bcast-1link") for k in order if (k, 1) in slice_bytes)
                s2 = sum(link_ms(slice_bytes[(k, 2)], W, "bcast-1link") for k in order if (k, 2) in slice_bytes)
                path1 = sum(x["chain_ms"] for x in ranks) + s1
                path2 = (sum(x["rows_behind_the_chain_ms"] + x["pack_ms"] for x in ranks) + big_link + unpack_all + s1 + s2) / 3.0
                t_chain = max(path1, path2)
                extra = {"slice1_chain_path_ms": path1, "panel_message_path_ms_over_3": path2}
            else:
                t_chain = sum(x["chain_ms"] + x["pack_ms"] for x in ranks) + big_link + unpack_all
                extra = {}
            lb = max(x["wall_ms"] for x in ranks)
            out["replays"].append({"world": W, "model": model, "ranks": ranks, "max_rank_wall_ms": lb, "owner_chain_path_ms": t_chain,
                                   "implied_step_ms": max(lb, t_chain), "implied_step_upper_ms": lb + t_chain, **extra,
                                   "all_owned_columns_bitwise_equal": all(x.get("owned_columns_bitwise_equal_to_recorded_factor", True)
                                                                          for x in ranks) if verify else None})
            print("[replay] W=%d %s: max rank wall %.1f ms, owner-chain path %.1f ms" % (W, model, lb, t_chain), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
