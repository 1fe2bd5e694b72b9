#!/usr/bin/env python
"""bench.py — Newton steps/sec at KKT dim 32k (BASELINE.json metric) on N GPUs of one node.

A "step" = one pass of the hot path over one synthetic dense QP with the derivative blocks already
resident in HBM: residual (K2) + KKT assembly (K1) + block-LDL' factorisation (K3/K4) +
substitutions (K5) + multiplier sign flip = /root/reference/pyipm.py:1717-1725 without
regularisation retries.  Workload at N=1: n=16384, me=4096, mi=6144 ->
KKT dim N = n + 2*mi + me = 32768.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 3 --warmup 1

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_update<128>, the fp64-MFMA
trailing update): algorithmic flops (sum over launches of 2*nb*#lower-triangle entries updated)
divided by the summed HIP-event durations of those launches, against the 78.6 TFLOP/s fp64 matrix
peak.  `cpu_baseline` times the oracle (the reference's CPU path restated) on this box's host cores
on a bounded sample and N^3-extrapolates to the metric's size.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # 256 CU x 2.4 GHz x 128 flop/clk/CU


def make_qp_device(n, me, mi, seed, device):
    """Device-side generator with the distribution of pyipm_amd.problems.make_qp;
    golden parity is checked at the sizes where the CPU oracle is feasible (tests/)."""
    import torch
    gen = torch.Generator(device=device).manual_seed(seed)
    f64 = torch.float64
    M = torch.randn(n, n, dtype=f64, device=device, generator=gen)
    Q = M @ M.T / n
    Q.diagonal().add_(1.0)
    del M
    A = torch.randn(me, n, dtype=f64, device=device, generator=gen) / np.sqrt(n)
    G = torch.randn(mi, n, dtype=f64, device=device, generator=gen) / np.sqrt(n)
    c = torch.randn(n, dtype=f64, device=device, generator=gen)
    x = torch.zeros(n, dtype=f64, device=device)
    s = torch.rand(mi, dtype=f64, device=device, generator=gen) * 1.5 + 0.5
    lam_i = torch.rand(mi, dtype=f64, device=device, generator=gen) * 1.5 + 0.5
    lam_e = torch.randn(me, dtype=f64, device=device, generator=gen)
    h = G @ x - s - 0.1 * torch.randn(mi, dtype=f64, device=device, generator=gen)
    b = A @ x - 0.1 * torch.randn(me, dtype=f64, device=device, generator=gen)
    return {"d2L": Q, "Je": A.T.contiguous(), "Ji": G.T.contiguous(), "df": Q @ x + c, "ce": A @ x - b,
            "ci": G @ x - h, "s": s, "lam": torch.cat([lam_e, lam_i]), "mu": 0.2}


def cpu_baseline(target_N=32768, eig_shape=(4096, 1024, 1536), lu_probe_shape=(6144, 1536, 2304), target_shape=(16384, 4096, 6144),
                 reps=3, lu_budget_s=170.0):
    """The oracle (reference CPU path restated: NumPy assembly + scipy eigvalsh(H, I) + scipy LU solve + flip =
    pyipm.py:1717-1725) timed on this box's host cores:
      * BLAS thread count: swept (a dense LU at N = 6144 per candidate), the fastest is used for everything below --
        all cores is NOT the fastest on a 2-socket box (OpenBLAS oversubscribes: round 1's figure suffered from that);
      * the rest of the step (assembly + scipy.linalg.solve(assume_a='gen') + flip) MEASURED ONCE AT THE METRIC'S OWN SIZE
        (N = 32768; round 5 -- rounds 1-4 extrapolated it from N = 12288): a probe at N = 12288 predicts its time first, and
        if that exceeds `lu_budget_s` (or the host lacks the memory: ~45 GB) the largest N that fits the budget is measured
        instead and N^3-extrapolated -- `legs` says which;
      * the reference's eigvalsh inertia test (one call of reghess) at N = 8192, median of `reps`, N^3-EXTRAPOLATED: the
        eigendecomposition is most of the reference's step and too slow to run at N = 32768 inside a benchmark that has to
        finish in minutes (~15 min; it grows faster than N^3 up to 8192 -- 3.4 s -> 14 s -- so the extrapolation flatters the CPU).
    A reported baseline, not the optimisation target."""
    from oracle import newton_oracle as orc
    from pyipm_amd.problems import make_qp
    import scipy.linalg
    ncpu = os.cpu_count() or 1
    blas, limits = "unknown", None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits as limits
        blas = ";".join(sorted(set("%s %s" % (i.get("internal_api"), i.get("version")) for i in threadpool_info())))
    except Exception:
        pass
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    avail = None
    try:
        import psutil
        avail = float(psutil.virtual_memory().available)
    except Exception:
        pass

    class _nolimit(object):
        def __init__(self, **kw): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    lim = limits if limits is not None else _nolimit
    cands = sorted(set(t for t in (8, 16, 32, 64, 128, ncpu) if 1 <= t <= ncpu))
    rng = np.random.default_rng(0)
    Mx = rng.standard_normal((6144, 6144))
    sweep = {}
    for t in cands:
        with lim(limits=t):
            scipy.linalg.lu_factor(Mx[:512, :512])
            t0 = time.perf_counter(); scipy.linalg.lu_factor(Mx, check_finite=False); sweep[t] = time.perf_counter() - t0
    del Mx
    best = min(sweep, key=sweep.get) if limits is not None else ncpu

    def median_of(fn, k):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            if ts[0] > 45.0:
                break
        return float(np.median(ts)), ts

    def problem(shape):
        n_, me_, mi_ = shape
        qp = make_qp(n_, me_, mi_, seed=0)
        return qp, (qp["d2L"], qp["Je"], qp["Ji"], qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], qp["mu"], n_, me_, mi_)

    dim = lambda sh: sh[0] + 2 * sh[2] + sh[1]              # noqa: E731
    with lim(limits=best):
        Ne, Np = dim(eig_shape), dim(lu_probe_shape)
        qe, _ = problem(eig_shape)
        He = orc.kkt_matrix(qe["d2L"], qe["Je"], qe["Ji"], qe["s"], qe["lam"], *eig_shape)
        t_eig, r_eig = median_of(lambda: orc.eigvalsh_ref(He), reps)      # the inertia test of reghess, one call (pyipm.py:1379)
        del He, qe
        _, al = problem(lu_probe_shape)
        t_probe, r_probe = median_of(lambda: orc.newton_step(*al, regularise=False), 1)      # assembly + LU + flip
        del al
        # the size the rest of the step is measured at: the metric's own if the probe says it fits the budget and the host
        # has the memory (the oracle's assembly holds ~4 copies of the N x N matrix, scipy's solve one more)
        sh = tuple(target_shape)
        scale = 1.0
        while True:
            Nl = dim(sh)
            fits_t = t_probe * (Nl / Np) ** 3 <= lu_budget_s
            fits_m = avail is None or 5.5 * 8.0 * Nl * Nl <= 0.8 * avail
            if (fits_t and fits_m) or Nl <= Np:
                break
            scale *= 0.75
            sh = tuple(int(round(v * scale / 128.0)) * 128 for v in target_shape)
        if Nl <= Np:
            sh, Nl, t_l_noeig, r_lu = lu_probe_shape, Np, t_probe, r_probe
        else:
            _, al = problem(sh)
            t_l_noeig, r_lu = median_of(lambda: orc.newton_step(*al, regularise=False), 1)
            del al
    lu_measured = Nl == target_N
    t_step_noeig = t_l_noeig * (target_N / Nl) ** 3
    t_step = t_eig * (target_N / Ne) ** 3 + t_step_noeig
    return {"value": 1.0 / t_step, "unit": "steps/s", "cores": int(best), "kind": "port",
            "sample": ("oracle/newton_oracle.py = pyipm.py:1717-1725 on the host (NumPy assembly + scipy.linalg.eigvalsh(H, I) "
                       "+ scipy.linalg.solve(assume_a='gen') + flip), %d BLAS threads (fastest of a sweep over %s on a dense LU "
                       "at N=6144: %s s) of %d host cores (%s; %s). Rest of the step (assembly + LU solve + flip) %s at N=%d "
                       "(n=%d,me=%d,mi=%d): %.1f s, once (probe at N=%d: %.2f s). eigvalsh(H, I) -- the inertia test of reghess, "
                       "one call -- at N=%d (n=%d,me=%d,mi=%d): %.2f s, median of %d, N^3-EXTRAPOLATED x%.0f to N=%d (on the MI355X "
                       "box's EPYC 9575F host it took 3.4 s at N=6144 and 14-15 s at N=8192: it grows faster than N^3 while its "
                       "tridiagonalisation leaves the caches, so the extrapolation flatters the CPU). value = 1 / (eigvalsh + rest)"
                       % (best, sorted(sweep), ", ".join("%.2f" % sweep[t] for t in sorted(sweep)), ncpu, cpu_model, blas,
                          "MEASURED at the metric's own size" if lu_measured else "measured at the largest size inside the time / memory budget and N^3-extrapolated x%.1f" % ((target_N / Nl) ** 3),
                          Nl, sh[0], sh[1], sh[2], t_l_noeig, Np, t_probe,
                          Ne, eig_shape[0], eig_shape[1], eig_shape[2], t_eig, len(r_eig), (target_N / Ne) ** 3, target_N)),
            "legs": {"rest_of_step(assembly+LU+flip)": {"N": Nl, "seconds": t_l_noeig, "kind": "measured" if lu_measured else "extrapolated",
                                                       "seconds_at_target_N": t_step_noeig},
                     "eigvalsh": {"N": Ne, "seconds": t_eig, "kind": "extrapolated", "seconds_at_target_N": t_eig * (target_N / Ne) ** 3}},
            "reps": len(r_eig), "runs_s": {"eigvalsh_N%d" % Ne: r_eig, "no_eigvalsh_N%d" % Nl: r_lu, "no_eigvalsh_probe_N%d" % Np: r_probe},
            "measured_N": Ne, "measured_N_without_eigvalsh": Nl, "threads_sweep_s": {str(k): v for k, v in sweep.items()},
            "measured_s_eigvalsh": t_eig, "measured_s_per_step_no_eigvalsh_at_N%d" % Nl: t_l_noeig,
            "value_no_eigvalsh": 1.0 / t_step_noeig}


def pmc_traffic(N, nb, bn=256):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md; tools/pmc_update.sh,
    tools/pmc_summary.py).  Counters cannot be collected inside this process, so the figure is attached
    only for the configuration it was measured on; otherwise null."""
    name = None
    for rnd in ("r06", "r05", "r04", "r03"):                    # the latest committed counter passes of this command
        cand = "%s_z_pmc_update.json" % rnd if bn == 256 else "%s_z_pmc_update_bn128.json" % rnd
        if os.path.exists(os.path.join(ROOT, "profiles", cand)):
            name = cand
            break
    if name is None:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    if N != 32768 or nb != 256 or os.environ.get("PYIPM_NEWTON_GROUP") not in (None, "8") or not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
        if ("<%d," % bn) not in d.get("kernel", ""):
            return None, None
        return float(d["hbm_bytes_per_launch_corrected"]), "profiles/%s (separate --pmc passes of this command)" % name
    except Exception:
        return None, None


def pmc_moved_fraction(N, nb):
    """HBM bytes actually moved / algorithmic bytes for the HBM-bound kernels (K1 k_assemble, the exposed backward sweep),
    from the latest committed counter passes of this command (tools/pmc_hbm.sh: FETCH_SIZE / WRITE_SIZE in separate runs).
    K1 leaves structural zeros in place and the sweeps skip the slack rows of x-block columns, so fewer bytes move than
    SURVEY 8d's algorithmic count: `achieved` on algorithmic bytes overstates the HBM rate by this factor."""
    if N != 32768 or nb != 256:
        return {}, None
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(ROOT, "profiles", "%s_z_pmc_hbm_kernels.json" % rnd)
        if os.path.exists(path):
            try:
                k = json.load(open(path))["kernels"]
                return ({"assemble_K1": k.get("k_assemble", {}).get("traffic_over_algorithmic"),
                         "solve_K5_exposed": k.get("k_bwd_sweep", {}).get("traffic_over_algorithmic")},
                        "profiles/%s_z_pmc_hbm_kernels.json (separate --pmc passes of this command)" % rnd)
            except Exception:
                return {}, None
    return {}, None


class Watchdog(object):
    """A multi-GPU run can stall where no Python code runs (RCCL bootstrap, the communicator's scatter + all-gather
    self-test, a collective a peer never joined).  The driver would then see no JSON at all.  Every phase that can stall
    is bracketed by watch(name, seconds); a daemon thread -- ctypes and torch release the GIL while they block -- prints ONE
    JSON line with an `error` field (rank 0; the other ranks only exit) and ends the process with status 3 when a phase
    overruns.  VERDICT r3 item 2(c)."""

    def __init__(self, json_fd, rank, base):
        import threading
        self.json_fd, self.rank, self.base = json_fd, rank, dict(base)
        self.phase, self.deadline, self.lock = None, None, threading.Lock()
        self.done = False
        self.fallback = None            # a complete result measured earlier in this run (the ladder's best completed rung)
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def watch(self, phase, seconds):
        with self.lock:
            self.phase, self.deadline = phase, (time.monotonic() + seconds) if seconds else None

    def clear(self):
        self.watch(None, None)

    def set_fallback(self, payload):
        with self.lock:
            self.fallback = dict(payload) if payload else None

    def fail_now(self, phase, reason):
        """A ladder rung that RAISES (an error code from the library: a poll of the tile chain timed out, PYIPM_E_COMM) is a rung
        that did not complete, like one that stalls: the best completed result stands; status 0 with it, 3 without."""
        with self.lock:
            fb = dict(self.fallback) if self.fallback else None
        if self.rank == 0:
            out = dict(self.base)
            if fb:
                out.update(fb)
                out.update({"ladder_failed_at": phase, "ladder_failure": str(reason)[:400],
                            "note": "a faster wire form failed; `value` is the best form that completed its timed steps"})
            else:
                out.update({"value": None, "error": "phase '%s' failed: %s" % (phase, str(reason)[:400]), "failed_phase": phase})
            try:
                os.write(self.json_fd, (json.dumps(out) + "\n").encode())
            except Exception:
                pass
        print("[bench] phase '%s' failed on rank %d (%s) -- %s" % (phase, self.rank, str(reason)[:200],
              "reporting the best completed wire form" if fb else "giving up"), file=sys.stderr, flush=True)
        os._exit(0 if fb else 3)

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            with self.lock:
                phase, dl = self.phase, self.deadline
            if dl is not None and time.monotonic() > dl:
                with self.lock:
                    fb = dict(self.fallback) if self.fallback else None
                if self.rank == 0:
                    out = dict(self.base)
                    if fb:
                        # a slower wire form of the ladder already measured this workload: that number stands
                        out.update(fb)
                        out.update({"ladder_stalled_at": phase,
                                    "note": "a faster wire form stalled (watchdog); `value` is the best form that completed its timed steps"})
                    else:
                        out.update({"value": None, "error": "watchdog: phase '%s' did not finish in time (stalled collective / "
                                                             "communicator bring-up?)" % phase, "stalled_phase": phase})
                    try:
                        os.write(self.json_fd, (json.dumps(out) + "\n").encode())
                    except Exception:
                        pass
                print("[bench] watchdog: phase '%s' overran on rank %d -- %s" % (phase, self.rank, "reporting the best completed wire form" if fb else "giving up"),
                      file=sys.stderr, flush=True)
                os._exit(0 if fb else 3)


# The wire forms of the distributed factorisation, SAFEST FIRST.  Across GPUs every rung is timed in turn --
# warm-up, then K steps between barriers -- and the headline region then runs the fastest rung that completed; from the second
# rung on the watchdog holds the best completed result, so a form that stalls (two communicators in flight on one device have
# never run on more than one GPU) costs its own number, not the run's: the line cannot come back without a value once the plain
# broadcast has been timed.  All rungs produce the same bits (tests/test_gpu_dist.py::test_exchange_forms_give_the_same_bits).
LADDER = (
    ("ncclBroadcast, one message per panel", {"dist_comm2": 0, "dist_slices": 0, "dist_sag": 0}),
    ("scatter + all-gather panels, one message per panel", {"dist_comm2": 0, "dist_slices": 0, "dist_sag": 1}),
    ("slices ahead of the panel message, one communicator", {"dist_comm2": 0, "dist_slices": 1, "dist_sag": 1}),
    ("slices ahead of the panel message, one communicator, the second slice's rows inside the chain's launch",
     {"dist_comm2": 0, "dist_slices": 2, "dist_sag": 1}),
    ("slices ahead of the panel message, slices on a second communicator", {"dist_slices": 2, "dist_sag": 1, "dist_comm2": 1}),
)


def run_ladder(core, one_step, fence, wd, world, rank, steps, warmup, reduce_max, base_payload, stall_rung=None):
    """Time every rung (safest first); returns (index of the fastest completed rung, [per-rung records])."""
    records = []
    best = None
    for i, (name, opts) in enumerate(LADDER):
        wd.watch("ladder rung %d: %s" % (i, name), float(os.environ.get("PYIPM_BENCH_LADDER_WATCH", 240 + 60 * (steps + warmup))))
        for k, v in opts.items():          # (dist_comm2 = 1 creates the second communicator here, collectively, on the RCCL transport)
            core.set_option(k, v)
        if stall_rung is not None and i == stall_rung:
            core.set_option("dist_timeout_s", 0.0)          # (test hook: this rung's first step stalls for 30 s and nothing bounds the wait)
            core.set_option("debug_fault", 3)
        try:
            for _ in range(max(1, warmup)):
                one_step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            fence()
            el = reduce_max(time.perf_counter() - t0)
        except Exception as e:          # (the other ranks stall in the rung's next collective: their watchdogs end them the same way)
            wd.fail_now("ladder rung %d: %s" % (i, name), e)
        rec = {"wire_form": name, "ms_per_step": 1e3 * el / steps, "value": steps / el, "steps": steps,
               "rccl_ranks": core.comm_ranks(), "panel_bcast": "scatter+allgather" if core.comm_bcast_mode() else "broadcast"}
        records.append(rec)
        if best is None or rec["value"] > records[best]["value"]:
            best = i
        fb = dict(base_payload)
        fb.update({"value": records[best]["value"], "ms_per_step": records[best]["ms_per_step"], "steps": steps,
                   "wire_form": records[best]["wire_form"], "rccl_ranks": records[best]["rccl_ranks"],
                   "panel_bcast": records[best]["panel_bcast"], "ladder": list(records)})
        wd.set_fallback(fb)
    for k, v in LADDER[best][1].items():
        core.set_option(k, v)
    return best, records


def self_launch(nproc):
    """Re-run this command line under torch.distributed.run with one rank per GPU on 127.0.0.1 (free port)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    rc = subprocess.call(cmd, env=env)
    if rc:
        raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nvar", "--n", dest="n", type=int, default=16384)
    ap.add_argument("--neq", "--me", dest="me", type=int, default=4096)
    ap.add_argument("--nineq", "--mi", dest="mi", type=int, default=6144)
    ap.add_argument("--nb", type=int, default=0, help="panel width (default 256 on one GPU; across GPUs 256 below KKT dimension 65536, where the owners' chain is the step, and 1024 from there on, where the bulk update is: tools/rank_replay.py, DESIGN.md section 6)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--refine", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="(kept for compatibility: the backward error is always reported)")
    ap.add_argument("--force-dist", action="store_true", help="use the per-panel distributed driver even for 1 GPU")
    ap.add_argument("--force-lookahead", action="store_true", help="with --force-dist --python-driver on one GPU: run the overlapped schedule")
    ap.add_argument("--python-driver", action="store_true", help="distributed runs: the Python loop over the per-panel phases instead of the library's driver")
    ap.add_argument("--selfmsg", action="store_true", help="with --force-dist on one GPU: pack and 'send' every panel anyway (message path cost)")
    ap.add_argument("--opt", action="append", default=[], help="core option name=value (e.g. tile_chain=0, lookahead=0)")
    ap.add_argument("--config4", choices=("auto", "on", "off"), default="auto",
                    help="after the timed region also time BASELINE.json's config 4 (n=65536, mi=32768 -> N=131072; 1 warm-up + 2 "
                         "steps) on the same GPUs and attach it as `config4`: the >= 5x-at-8-GPUs target is stated on THAT size, so "
                         "every 1-GPU and N-GPU line carries its leg of it.  auto = only with the default headline workload")
    ap.add_argument("--configs", choices=("auto", "on", "off"), default="auto",
                    help="after the timed region also time BASELINE.json's configs 2, 3 and 5 (n=2048/mi=2048; n=16384/me=mi=8192; 512 "
                         "batched n=256 QPs) on one GPU and attach them as `config2` / `config3` / `config5`.  auto = only with the "
                         "default headline workload on one GPU")
    ap.add_argument("--no-clock", action="store_true",
                    help="skip the extra step that measures the shader clock of the update kernel (counter-collection runs: the "
                         "traced process then holds exactly warmup + steps steps)")
    ap.add_argument("--ladder", choices=("auto", "on", "off"), default="auto",
                    help="across GPUs: time every wire form of the distributed factorisation, safest first, and run the headline region "
                         "on the fastest one that completed (auto = with more than one rank and the library's driver)")
    ap.add_argument("--extras", action="store_true",
                    help="after the timed region also run and report (a) the all-dense factorisation (skip_zeros=0) with a "
                         "bitwise check of the direction and (b) one L-BFGS search direction (SURVEY 8f rank 4).  Off by "
                         "default so that a kernel trace of the default command holds the headline workload only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one process per GPU ourselves (the same command line
        # the driver uses) and hand its single JSON line through.  VERDICT r2: this used to exit before measuring.
        return self_launch(args.gpus)

    # stdout carries exactly ONE line (the JSON): native libraries print there too (RCCL's version banner on
    # rank 0), so fd 1 is pointed at stderr for the whole run and the JSON goes to a private copy of it
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pyipm_amd.newton import NewtonCore, mfma_f64_peak

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_gpu_dist.py): several ranks on ONE GPU over gloo -- exercises this file's multi-rank path
    # (row-sharded staging, the library's distributed driver with callback exchange, the JSON line) on a one-GPU box;
    # RCCL itself refuses two ranks on one device.  Never set by the driver.
    share_gpu = os.environ.get("PYIPM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if args.gpus != world and not (world == 1 and args.gpus > 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Newton-step core has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    global _WD
    wd = _WD = Watchdog(json_fd, rank, {"metric": "newton_steps_per_sec", "unit": "steps/s", "n_gpus": world, "steps": args.steps,
                                  "warmup": args.warmup, "higher_is_better": True, "scaling": "strong", "dtype": "f64",
                                  "data": "synthetic"})
    wd.watch("process-group initialisation", 600)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    # across GPUs the panel width trades the owners' chain (proportional to nb) against the efficiency of the bulk update
    # (rank-nb) and the number of messages: the rank replay on one GPU (profiles/r04_z_replay_*.json, r04_z_replay_nb.txt) puts
    # N = 32768 at 47 / 55 / 55 ms for nb = 256 / 512 / 1024 on 8 ranks (90 / 106 / 114 on 2) and N = 131072 at 0.71 s for
    # nb = 1024 against 0.81 s for 512.  Chosen per workload (the config-4 leg has its own).
    nb_user = args.nb

    def pick_nb(kkt_dim):
        return nb_user if nb_user else default_panel_width(world, kkt_dim)
    args.nb = pick_nb(args.n + 2 * args.mi + args.me)
    n, me, mi = args.n, args.me, args.mi
    N = n + 2 * mi + me
    want_condensed = any(kv.split("=")[0] == "condensed" and float(kv.split("=")[1]) != 0 for kv in args.opt) and mi > 0
    condensed = want_condensed

    def build(n, me, mi, seed):
        """Problem + handle + the step callable for one workload (the headline one, then config 4)."""
        qp = make_qp_device(n, me, mi, seed, device)
        core = NewtonCore(n, me, mi, device=local_rank, nb=pick_nb(n + 2 * mi + me), world=world, rank=rank)
        if world > 1 and want_condensed:
            # the condensed option across ranks takes the full blocks on every rank (a column of Ji Sigma Ji' needs every row of Ji)
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        elif world > 1:
            # row-sharded staging: a rank assembles only the KKT columns it owns, i.e. it needs only those rows of
            # d2L / Je / Ji (every rank generated the same matrices from the same seed; the full copies are dropped)
            rows = torch.from_numpy(core.owned_rows()).to(device)
            core.stage_blocks_owned(qp["d2L"].index_select(0, rows), qp["Je"].index_select(0, rows) if me else None,
                                    qp["Ji"].index_select(0, rows) if mi else None)
            qp["d2L"] = qp["Je"] = qp["Ji"] = None
            torch.cuda.empty_cache()
        else:
            core.stage_blocks(qp["d2L"], qp["Je"], qp["Ji"])
        core.stage_vectors(qp["df"], qp["ce"], qp["ci"], qp["s"], qp["lam"], mu=qp["mu"])
        core.set_option("profile", 1)
        core.set_option("expert", 1)                         # (--opt may name an expert switch; the clock stamps are one)
        for kv in args.opt:
            k, v = kv.split("=")
            core.set_option(k, float(v))
        if use_dist:
            # the library's own per-panel driver (pyipm_newton_step_dist); exchange = a handle-owned RCCL communicator
            from pyipm_amd.dist import DistNewton
            wd.watch("communicator bring-up (ncclCommInitRank + scatter/all-gather self-test)", 600)
            drv = DistNewton(core, native=not args.python_driver)
            drv.force_lookahead = args.force_lookahead
            if args.selfmsg and world == 1:
                core.set_option("dist_selfmsg", 1)
            return qp, core, (lambda: drv.step(0.0, 0.0, refine=args.refine))
        return qp, core, (lambda: core.step(0.0, 0.0, refine=args.refine))

    wd.watch("problem generation + staging", 900)
    qp, core, one_step = build(n, me, mi, args.seed)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ladder = None
    if use_dist and not args.python_driver and (args.ladder == "on" or (args.ladder == "auto" and world > 1)) and \
            not any(kv.split("=")[0] in ("dist_slices", "dist_sag", "dist_comm2") for kv in args.opt):
        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if share_gpu else device)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        base_payload = {"vs_baseline": None,
                        "config": {"workload": "synthetic convex dense QP Newton step (residual+KKT assembly+block LDL^T+solve+flip), "
                                               "n=%d me=%d mi=%d -> KKT dim %d" % (n, me, mi, N), "kkt_dim": N, "n": n, "me": me, "mi": mi,
                                   "nb": core.nb, "parallelism": "1D block-cyclic column panels over %d GPU(s)" % world}}
        stall = os.environ.get("PYIPM_BENCH_LADDER_STALL")          # test hook (tests/test_gpu_dist.py): that rung never completes
        best, recs = run_ladder(core, one_step, fence, wd, world, rank, max(2, min(args.steps, 5)), 1, reduce_max, base_payload,
                                stall_rung=int(stall) if stall else None)
        ladder = {"chosen": recs[best]["wire_form"], "rungs": recs}

    wd.watch("warm-up steps", 300 + 120 * args.warmup)
    for _ in range(args.warmup):
        one_step()
    trailing_ms = trailing_flops = panel_ms = solve_ms = assemble_ms = gram_ms = trailing_area = 0.0
    n_launch = 0
    inst = {128: {"launches": 0, "ms": 0.0, "flops": 0.0, "area": 0.0}, 256: {"launches": 0, "ms": 0.0, "flops": 0.0, "area": 0.0}}
    dist_ms = {}
    inst_bytes = {128: {"c_tiles": 0.0, "c_tiles_and_panels": 0.0}, 256: {"c_tiles": 0.0, "c_tiles_and_panels": 0.0}}
    wd.watch("timed steps", 300 + 120 * args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dz, st = one_step()
        tm = core.timings()         # reads HIP-event durations of this step (stream already drained by the step)
        trailing_ms += tm["trailing_ms"]; trailing_flops += tm["trailing_flops"]; n_launch += tm["n_trailing"]
        panel_ms += tm["panel_ms"]; solve_ms += tm["solve_ms"]; assemble_ms += tm["assemble_ms"]
        gram_ms += tm["gram_ms"]
        trailing_area += tm["trailing_area"]
        for bn, v in core.trailing_instances().items():      # the bulk launches by kernel instance (128 x 128 / 128 x 256 tiles)
            for k2 in v:
                inst[bn][k2] += v[k2]
        for bn, v in core.trailing_bytes().items():
            for k2 in v:
                inst_bytes[bn][k2] += v[k2]
        if use_dist and not args.python_driver:
            for k, v in core.dist_timings().items():
                dist_ms[k] = dist_ms.get(k, 0.0) + v
    fence()
    elapsed = time.perf_counter() - t0
    wd.watch("backward-error check / reductions after the timed region", 600)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of the timed steps' direction, outside the timed region: |Hc dz - g| / |g| with Hc applied from the KKT
    # blocks (never from the factor); dz is the last timed step's output with the flip undone.  (All ranks take part.)
    raw = dz.clone()
    if me + mi:
        raw[n + mi:] *= -1.0
    if world > 1:
        g_res = core.residual_dist()
        berr = float((core.matvec_dist(raw) - g_res).norm() / g_res.norm())
    else:
        g_res = core.residual()
        berr = float((core.matvec(raw) - g_res).norm() / g_res.norm())

    # The shader clock the bulk update actually runs at (round 4): one more step with the update kernel's per-block stamps on
    # (diagnostics buffer: 100 MHz wall clock and shader cycle counter around each block's main loop).  78.6 TFLOP/s is 256 CUs x
    # 128 flop/clk at 2.4 GHz; under this load the part clocks lower, and cycles / time says by how much.
    clock = None
    if world == 1 and not use_dist and not condensed and not args.no_clock:
        try:
            nrec = (core.Npad // 128) ** 2 + 4096
            tl = torch.zeros(nrec * 8, dtype=torch.int64, device=device)
            core.set_option("debug_timeline_ptr", float(tl.data_ptr()))
            one_step(); torch.cuda.synchronize()
            core.set_option("debug_timeline_ptr", 0.0)
            rec = tl.cpu().numpy().reshape(-1, 8)
            rec = rec[rec[:, 0] != 0]
            loop_us = (rec[:, 2] - rec[:, 1]) * 0.01
            cyc = rec[:, 4] >> 16
            ok = loop_us > 20.0
            mhz = cyc[ok] / loop_us[ok]
            if mhz.size:
                p10, p50, p90 = (float(v) for v in np.percentile(mhz, [10, 50, 90]))
                clock = {"shader_mhz_in_update_main_loop": {"p10": p10, "median": p50, "p90": p90, "blocks": int(mhz.size)},
                         "fp64_mfma_peak_at_that_clock_tflops": 256 * 128 * p50 * 1e6 / 1e12,
                         "method": "one extra step after the timed region with the update kernel's per-block stamps on: "
                                   "clock64() ticks over s_memrealtime (100 MHz) time of each block's main loop"}
        except Exception as e:          # diagnostics only
            clock = {"error": str(e)}

    # ranks of the handle-owned RCCL communicator as RCCL itself counts them (0 on one GPU: no exchange exists)
    rccl_ranks = core.comm_ranks() if use_dist else 0
    if rank == 0:
        K = args.steps
        ach_all = (trailing_flops / 1e12) / (trailing_ms * 1e-3) if trailing_ms > 0 else 0.0
        # the dominant kernel = the k_update instance that executed most of the bulk flops (128 x 256 tiles since round 3; the
        # chain-bound phase and K < 512 launches keep 128 x 128): the roofline object is ITS launches only, so that a kernel
        # trace's per-name average agrees; the other instance gets the same figures in `other_instance`
        dom = 256 if inst[256]["flops"] >= inst[128]["flops"] else 128
        oth = 128 if dom == 256 else 256
        if inst[dom]["launches"] == 0:                       # (python driver / no profile: fall back to the sums)
            inst[dom] = {"launches": n_launch, "ms": trailing_ms, "flops": trailing_flops, "area": 16.0 * trailing_area}

        def inst_obj(bn):
            v = inst[bn]
            a_ = (v["flops"] / 1e12) / (v["ms"] * 1e-3) if v["ms"] > 0 else 0.0
            ok = bool(v["launches"] and world == 1 and not condensed)
            traffic = pmc_traffic(N, args.nb, bn)[0]
            return {"kernel": "k_update<%d,true,8> (fp64 MFMA trailing rank-K update, 128 x %d tiles)" % (bn, bn),
                    "achieved": a_, "frac": a_ / FP64_MFMA_PEAK_TFLOPS, "launches": v["launches"],
                    "avg_launch_ms": v["ms"] / max(v["launches"], 1), "flops_per_launch_avg": v["flops"] / max(v["launches"], 1),
                    # both definitions, so that rounds stay comparable: round 2 counted the C tiles only,
                    # since round 3 the two operand panels (read once) count as well
                    "algorithmic_bytes_per_launch": (v["area"] / v["launches"]) if ok else None,
                    "algorithmic_bytes_per_launch_c_tiles_only": (inst_bytes[bn]["c_tiles"] / v["launches"]) if ok else None,
                    "algorithmic_bytes_per_launch_c_tiles_and_panels": (inst_bytes[bn]["c_tiles_and_panels"] / v["launches"]) if ok else None,
                    # the invariant across definitions: HBM-side bytes (PMC) per thousand flops of the launch
                    "bytes_per_kflop": (traffic / (v["flops"] / v["launches"]) * 1e3) if (traffic and v["launches"] and v["flops"]) else None,
                    "algorithmic_bytes_per_kflop": (v["area"] / v["flops"] * 1e3) if (ok and v["flops"]) else None}
        dobj, oobj = inst_obj(dom), inst_obj(oth)
        ach = dobj["achieved"]
        try:
            peak_meas = mfma_f64_peak(local_rank, 20000)
        except Exception:
            peak_meas = None
        out = {
            "metric": "newton_steps_per_sec", "value": K / elapsed, "unit": "steps/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "panel_bcast": ("scatter+allgather" if (use_dist and core.comm_bcast_mode()) else ("ncclBroadcast" if rccl_ranks else None)),
            "wire_form": ladder["chosen"] if ladder else None,
            "ladder": ladder["rungs"] if ladder else None,
            "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic convex dense QP Newton step (residual+KKT assembly+block LDL^T+solve+flip), "
                                   "n=%d me=%d mi=%d -> KKT dim N=%d, seed %d, nb=%d, refine=%d"
                                   % (n, me, mi, N, args.seed, args.nb, args.refine),
                       "kkt_dim": N, "n": n, "me": me, "mi": mi, "nb": args.nb,
                       "parallelism": "1D block-cyclic column panels over %d GPU(s)" % world,
                       "pivoting": "Bunch-Kaufman restricted to 64x64 diagonal tiles (block pivots); static pivots + refinement where a tile cannot pivot on its own"},
            "roofline": {"bound": "mfma", "kernel": dobj["kernel"],
                         "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(N, args.nb, dom)[0],
                         "traffic_unit": "bytes per launch (HBM side, PMC)", "traffic_source": pmc_traffic(N, args.nb, dom)[1],
                         "algorithmic_bytes_per_launch": dobj["algorithmic_bytes_per_launch"],
                         "algorithmic_bytes_per_launch_c_tiles_only": dobj["algorithmic_bytes_per_launch_c_tiles_only"],
                         "algorithmic_bytes_per_launch_c_tiles_and_panels": dobj["algorithmic_bytes_per_launch_c_tiles_and_panels"],
                         "algorithmic_bytes_note": "c_tiles_only = 16 B per updated entry (round 2's definition); c_tiles_and_panels "
                                                   "adds the two operand panels read once (the definition of `algorithmic_bytes_per_launch` "
                                                   "since round 3); bytes_per_kflop = traffic / flops of a launch is the same under both",
                         "bytes_per_kflop": dobj["bytes_per_kflop"], "algorithmic_bytes_per_kflop": dobj["algorithmic_bytes_per_kflop"],
                         "launches": dobj["launches"], "avg_launch_ms": dobj["avg_launch_ms"],
                         "flops_per_launch_avg": dobj["flops_per_launch_avg"],
                         "peak_measured_mfma_only": peak_meas,
                         "frac_of_measured_peak": (ach / peak_meas) if peak_meas else None,
                         "clock": clock,
                         "frac_of_peak_at_measured_clock": (ach / clock["fp64_mfma_peak_at_that_clock_tflops"])
                                                           if (clock and "fp64_mfma_peak_at_that_clock_tflops" in clock) else None,
                         "all_bulk_launches": {"achieved": ach_all, "launches": n_launch, "avg_launch_ms": trailing_ms / max(n_launch, 1)},
                         "other_instance": dict(oobj, traffic=pmc_traffic(N, args.nb, oth)[0]) if oobj["launches"] else None},
            "phases_ms_per_step": {"assemble": assemble_ms / K, "panel(tile+scale+in-panel)": panel_ms / K,
                                   "trailing": trailing_ms / K, "solve": solve_ms / K},
            "phases_note": "trailing = sum of the bulk update launches' HIP-event durations on the main stream (what a kernel "
                           "trace adds up for k_update<256,true,8> and k_update<128,true,8> together); panel = the rest of the factorisation: the tile chain where "
                           "no bulk launch runs, including the lookahead heads, which ride the chain's stream as "
                           "k_update<128,true,4> and are not part of the roofline figures",
            # the HBM-bound kernels, one roofline object each (bound, achieved, peak, frac): K1 = k_assemble on SURVEY 8d's
            # algorithmic bytes; K5 = the exposed part of the substitutions (block-diagonal + backward sweep: the factor
            # read once, 4 N^2 B; the forward sweep is hidden under the factorisation)
            "hbm_bound_kernels": {
                "assemble_K1": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                                "algorithmic_bytes": 4.0 * N * N + 8.0 * (n * n / 2.0 + n * me + n * mi),
                                "achieved": (4.0 * N * N + 8.0 * (n * n / 2.0 + n * me + n * mi)) / max(assemble_ms / K, 1e-9) / 1e6,
                                "frac": (4.0 * N * N + 8.0 * (n * n / 2.0 + n * me + n * mi)) / max(assemble_ms / K, 1e-9) / 1e6 / 8000.0,
                                "ms": assemble_ms / K},
                "solve_K5_exposed": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                                     "note": "forward pass is fused under the factorisation; exposed part = block-diagonal + "
                                             "backward sweep: lower triangle of the factor read once",
                                     "algorithmic_bytes": 4.0 * N * N, "algorithmic_bytes_total_solve": 8.0 * N * N,
                                     "achieved": 4.0 * N * N / max(solve_ms / K, 1e-9) / 1e6,
                                     "frac": 4.0 * N * N / max(solve_ms / K, 1e-9) / 1e6 / 8000.0, "ms": solve_ms / K},
            },
            # dense count of SURVEY.md 8d (what a dense LDL' of the reference's matrix costs).  The factorisation skips
            # the tiles the KKT block structure leaves at exact zero, so the dense-equivalent rate can exceed the
            # MFMA peak; "executed" counts the flops of the trailing-update launches actually issued.
            "step_flops_algorithmic": N ** 3 / 3.0 + 2.0 * N ** 2,
            "step_tflops": (N ** 3 / 3.0 + 2.0 * N ** 2) / (elapsed / K) / 1e12,
            "step_frac_of_peak": (N ** 3 / 3.0 + 2.0 * N ** 2) / (elapsed / K) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
            "step_flops_note": "dense-equivalent N^3/3 + 2N^2; structural zeros of the KKT matrix are skipped",
            "trailing_flops_executed_per_step": trailing_flops / K,
            "trailing_executed_over_dense": (trailing_flops / K) / (N ** 3 / 3.0),
            "step_frac_of_peak_executed_trailing": (trailing_flops / K) / (elapsed / K) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
            "inertia": {"n_neg": st["n_neg"], "expected": me + mi, "n_zero": st["n_zero"], "n_2x2": st["n_2x2"],
                        "growth": st["growth"]},
        }
        # moved-bytes fraction beside the algorithmic one: what the counters say really crosses the HBM side
        mf, mf_src = pmc_moved_fraction(N, args.nb)
        for key, frac in mf.items():
            hk = out["hbm_bound_kernels"].get(key)
            if hk is not None and frac:
                hk["moved_bytes_over_algorithmic"] = frac
                hk["achieved_on_moved_bytes"] = hk["achieved"] * frac
                hk["frac_on_moved_bytes"] = hk["achieved"] * frac / 8000.0
                hk["moved_bytes_source"] = mf_src
        # (algorithmic bytes per launch: the C tiles read and written once, 16 B per matrix entry a launch updates, + the
        # two operand panels read once -- 37 % on top at K = 2048)
        if dist_ms:
            # rank 0's view of the distributed schedule, per step: wall time of the factorisation, its own panel
            # factorisations (chain), packing, broadcasts as seen on the collective stream, rebuilding L from received
            # panels (unpack), the sweeps; bytes / messages of the panel exchange
            out["dist_phases_per_step"] = {k: v / K for k, v in dist_ms.items()}
            out["dist_driver"] = "pyipm_newton_step_dist (per-panel schedule in C); exchange: %s" % (
                "callbacks over gloo (test hook: ranks share a GPU)" if share_gpu else "handle-owned RCCL communicator" if world > 1 else "none (one rank)")
        if condensed:
            # same Newton direction from the (n+me)-dimensional condensed system; NOT the
            # headline configuration: the flop count of the step itself changes
            Nc = n + me
            fl = Nc ** 3 / 3.0 + float(mi) * n * n + 2.0 * Nc ** 2 + 4.0 * n * mi
            out["config"]["kkt_form"] = "condensed: [[H + Ji Sigma Ji', Je],[Je', 0]] of dimension %d (s, lambda_i eliminated)" % Nc
            out["phases_ms_per_step"]["gram(Ji Sigma Ji' MFMA launch, inside assemble)"] = gram_ms / K
            out["gram_tflops"] = (float(mi) * n * n / 1e12) / max(gram_ms / K * 1e-3, 1e-12)
            out["step_flops_algorithmic"] = fl
            out["step_tflops"] = fl / (elapsed / K) / 1e12
            out["step_frac_of_peak"] = out["step_tflops"] / (FP64_MFMA_PEAK_TFLOPS * world)
            out["roofline"]["algorithmic_bytes_per_launch"] = None
            out["hbm_bound_kernels"].pop("assemble_K1", None)
        else:
            out["config"]["kkt_form"] = "full 4-block system of the reference (pyipm.py:816-844)"
        if args.extras and world == 1 and not use_dist and not condensed and not any(kv.startswith("skip_zeros") for kv in args.opt):
            # transparency: the same step with the structural-zero skipping switched off (all-dense factorisation,
            # bitwise the same direction); not part of `value`
            core.set_option("skip_zeros", 0)
            core.step(0.0, 0.0)
            torch.cuda.synchronize()
            td = time.perf_counter()
            for _ in range(2):
                dzd, _std = core.step(0.0, 0.0)
            torch.cuda.synchronize()
            td = (time.perf_counter() - td) / 2
            core.set_option("skip_zeros", 1)
            dz1, _ = core.step(0.0, 0.0)
            out["all_dense_factorisation"] = {"value": 1.0 / td, "unit": "steps/s", "ms_per_step": 1e3 * td,
                                              "same_direction_bitwise": bool(torch.equal(dzd, dz1)),
                                              "note": "skip_zeros=0: every tile of the dense N^3/3 is computed; "
                                                      "`value` skips tiles the KKT block pattern makes exact zeros"}
        if args.extras and world == 1 and not use_dist:
            out["lbfgs_direction"] = lbfgs_block(device)
        out["backward_error"] = berr
        out["backward_error_note"] = "|Hc dz - g|/|g| of the last timed step, Hc from the staged blocks (pyipm_newton_kkt_matvec)"

    # ---- the other BASELINE.json configurations, outside the timed region of `value` ------------------------------------
    default_workload = (n, me, mi) == (16384, 4096, 6144) and not args.opt and not args.force_dist and not args.python_driver
    want_c4 = args.config4 == "on" or (args.config4 == "auto" and default_workload)
    want_legs = args.configs == "on" or (args.configs == "auto" and default_workload and world == 1)
    if want_c4 or want_legs:
        core.close()
        one_step = None                                      # (the closure holds the handle and its workspace)
        del core, qp, dz, raw, g_res
        torch.cuda.empty_cache()
    if want_legs and world == 1:
        # configs[1] (n = 2048, mi = 2048: chain-bound), configs[2] (n = 16384, me = mi = 8192 -> N = 40960: the MFMA roofline
        # run) and configs[4] (512 independent n = 256 QPs, batched handle): every BASELINE config is driver-timed
        wd.watch("config 2 / 3 / 5 legs", 900)
        out["config2"] = single_gpu_leg(build, "BASELINE.json configs[1]", 2048, 0, 2048, steps=20, warmup=3)
        out["config3"] = single_gpu_leg(build, "BASELINE.json configs[2]", 16384, 8192, 8192, steps=2, warmup=1)
        out["config5"] = batched_leg(device)
    if want_c4:
        c4 = config4_leg(build, fence, wd, world, rank, device, use_dist, share_gpu)
        if rank == 0:
            out["config4"] = c4
    if rank == 0:
        wd.watch("cpu baseline", 1800)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(target_N=N, target_shape=(n, me, mi))
        wd.clear()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    wd.watch("process-group teardown", 300)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    wd.done = True


def default_panel_width(world, kkt_dim):
    """Panel width when --nb is not given: 256 on one GPU; across GPUs 256 while the owners' chain is the step (KKT dimension
    below 65536) and 1024 where the bulk update is (tools/rank_replay.py, profiles/r04_z_replay_nb.txt)."""
    return 256 if (world == 1 or kkt_dim < 65536) else 1024


def single_gpu_leg(build, label, n, me, mi, steps, warmup):
    """One more BASELINE.json configuration on ONE GPU through the same build() / step as the headline: ms per step (wall,
    synchronised on both sides), backward error from the blocks, inertia, the bulk update kernels' rate and share, and the
    share of the step during which no bulk launch runs (the exposed tile chain: what bounds a small system)."""
    import torch
    N = n + 2 * mi + me
    qp, core, one_step = build(n, me, mi, 0)
    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {"trailing_ms": 0.0, "trailing_flops": 0.0, "panel_ms": 0.0, "assemble_ms": 0.0, "solve_ms": 0.0, "factor_ms": 0.0}
    inst = {128: {"ms": 0.0, "flops": 0.0, "launches": 0}, 256: {"ms": 0.0, "flops": 0.0, "launches": 0}}
    for _ in range(steps):
        dz, st = one_step()
        tm = core.timings()
        for k in acc:
            acc[k] += tm[k]
        for bn, v in core.trailing_instances().items():
            for k in inst[bn]:
                inst[bn][k] += v[k]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    raw = dz.clone()
    if me + mi:
        raw[n + mi:] *= -1.0
    g_res = core.residual()
    berr = float((core.matvec(raw) - g_res).norm() / g_res.norm())
    ms = 1e3 * el / steps
    dom = 256 if inst[256]["flops"] >= inst[128]["flops"] else 128
    dk = inst[dom]
    ach = (dk["flops"] / 1e12) / (dk["ms"] * 1e-3) if dk["ms"] > 0 else 0.0
    out = {"workload": "%s: synthetic convex dense QP Newton step, n=%d me=%d mi=%d -> KKT dim N=%d, seed 0, nb=%d" % (label, n, me, mi, N, core.nb),
           "kkt_dim": N, "steps": steps, "warmup": warmup, "ms_per_step": ms, "value": steps / el, "unit": "steps/s",
           "backward_error": berr, "inertia": {"n_neg": st["n_neg"], "expected": me + mi, "n_zero": st["n_zero"], "n_2x2": st["n_2x2"]},
           "phases_ms_per_step": {"assemble": acc["assemble_ms"] / steps, "factor": acc["factor_ms"] / steps,
                                  "panel(no bulk launch running)": acc["panel_ms"] / steps, "trailing(sum of bulk launches)": acc["trailing_ms"] / steps,
                                  "solve": acc["solve_ms"] / steps},
           "chain_share_of_step": (acc["panel_ms"] / steps) / ms,
           "dominant_kernel": {"kernel": "k_update<%d,true,8>" % dom, "achieved": ach, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                               "launches": dk["launches"], "avg_launch_ms": dk["ms"] / max(dk["launches"], 1),
                               "share_of_step_time": (dk["ms"] / steps) / ms},
           "all_bulk_launches_tflops": (acc["trailing_flops"] / 1e12) / (acc["trailing_ms"] * 1e-3) if acc["trailing_ms"] > 0 else None,
           "step_tflops_dense_equivalent": (N ** 3 / 3.0 + 2.0 * N ** 2) / (el / steps) / 1e12,
           "step_frac_of_peak_executed_trailing": (acc["trailing_flops"] / steps) / (el / steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    core.close()
    del core, qp
    torch.cuda.empty_cache()
    return out


def batched_leg(device, B=512, n=256, me=0, mi=256, steps=5, warmup=2):
    """BASELINE.json configs[4]: 512 independent n = 256 QPs (N = 768 each) through the batched handle, one GPU ("replicas
    only": across GPUs the batch is split by rank, no exchange).  Both forms are timed: the full 4-block system of the
    reference and the condensed one ("condensed" = 1: s and lambda_i eliminated per problem, n + me = 256 columns factored;
    the same directions, checked here against the blocks).  Rates: dense-equivalent flops of the FULL system per second (what
    the reference's LU would spend) and bytes moved per second (the blocks read once + the direction written: the floor of
    any implementation)."""
    import torch
    from pyipm_amd.batched import BatchedNewton
    f64 = torch.float64
    N = n + 2 * mi + me
    gen = torch.Generator(device=device).manual_seed(0)
    M = torch.randn(B, n, n, dtype=f64, device=device, generator=gen)
    Q = M @ M.transpose(1, 2) / n + torch.eye(n, dtype=f64, device=device)
    del M
    G = torch.randn(B, mi, n, dtype=f64, device=device, generator=gen) / n ** 0.5
    Ji = G.transpose(1, 2).contiguous()
    df = torch.randn(B, n, dtype=f64, device=device, generator=gen)
    s = torch.rand(B, mi, dtype=f64, device=device, generator=gen) * 1.5 + 0.5
    lam = torch.rand(B, mi, dtype=f64, device=device, generator=gen) * 1.5 + 0.5
    ci = s + 0.1 * torch.randn(B, mi, dtype=f64, device=device, generator=gen)
    mu, eps = 0.2, float(np.finfo(np.float64).eps)
    # right-hand side and the product Hc dz from the blocks (torch: outside every timed region, checking only)
    g = torch.cat([-(df - torch.bmm(Ji, lam.unsqueeze(2)).squeeze(2)), -(lam - mu / (s + eps)), -(ci - s)], dim=1)

    def backward_error(dz):
        dx, ds, dl = dz[:, :n], dz[:, n:n + mi], -dz[:, n + mi:]
        Qs = torch.triu(Q) + torch.triu(Q, 1).transpose(1, 2)
        r = torch.cat([torch.bmm(Qs, dx.unsqueeze(2)).squeeze(2) + torch.bmm(Ji, dl.unsqueeze(2)).squeeze(2),
                       lam / (s + eps) * ds - dl, torch.bmm(Ji.transpose(1, 2), dx.unsqueeze(2)).squeeze(2) - ds], dim=1)
        return float(((r - g).norm(dim=1) / g.norm(dim=1)).max())

    flops_dense = B * (N ** 3 / 3.0 + 2.0 * N ** 2)
    bytes_min = 8.0 * B * (n * (n + 1) / 2.0 + n * mi + n + 3 * mi + (me + mi) + N)
    out = {"workload": "BASELINE.json configs[4]: %d independent synthetic convex QPs, n=%d me=%d mi=%d -> N=%d each, seed 0" % (B, n, me, mi, N),
           "batch": B, "kkt_dim": N, "steps": steps, "warmup": warmup, "forms": {}}
    for form, cond in (("full", 0), ("condensed", 1)):
        bn = BatchedNewton(n, me, mi, condensed=bool(cond), guard=False)     # (the timed loop is the bare step; the guard's check follows)
        for _ in range(warmup):
            dz, st = bn.step_all(Q, None, Ji, df, None, ci, s, lam, mu=mu)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            dz, st = bn.step_all(Q, None, Ji, df, None, ci, s, lam, mu=mu)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        # flops the form EXECUTES (dense count of what it factors, structure of the slack block not subtracted): the full form N^3/3 + 2 N^2
        # per problem; the condensed one the Gram part 2 n^2 mi (lower block triangle: x 10/16 at n = 256) + F_step(n + me)
        nc = n + me
        executed = B * ((2.0 * n * n * mi * (10.0 / 16.0 if n == 256 else 0.5 + 0.5 * 64.0 / max(n, 64)) + nc ** 3 / 3.0 + 2.0 * nc ** 2) if cond
                        else (N ** 3 / 3.0 + 2.0 * N ** 2))
        kms = bn.last_ms()
        out["forms"][form] = {"ms_per_batch_step": 1e3 * el, "value": B / el, "unit": "Newton steps/s (problems x steps)",
                              "executed_flops": executed, "executed_tflops": executed / el / 1e12,
                              "executed_frac_of_mfma_peak": executed / el / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                              "reference_lu_equivalent_tflops_NOT_a_roofline_fraction": flops_dense / el / 1e12,
                              "wall_minus_kernels_ms": 1e3 * el - kms.get("step_ms", 0.0),
                              "bytes_moved_floor_gbs": bytes_min / el / 1e9, "bytes_moved_floor_frac_of_hbm": bytes_min / el / 1e9 / 8000.0,
                              "backward_error_max": backward_error(dz),
                              "inertia_ok": bool(all(x["n_neg"] == me + mi and x["n_zero"] == 0 for x in st)),
                              "backward_error_max_device_check": float(bn.backward_errors(dz).max()),
                              "kernel_ms": kms}
        bn.close()
    best = min((v for v in out["forms"].values() if "ms_per_batch_step" in v), key=lambda v: v["ms_per_batch_step"])
    out["ms_per_batch_step"] = best["ms_per_batch_step"]
    out["value"] = best["value"]; out["unit"] = best["unit"]
    out["roofline_note"] = ("one workgroup per problem: a chain of dependent tile inversions and block solves -- latency-bound, at neither "
                            "roofline (executed_tflops / executed_frac_of_mfma_peak: the flops the form executes); "
                            "reference_lu_equivalent_tflops counts what the reference's LU of the FULL system would spend per problem "
                            "(N^3/3 + 2N^2: the condensed form executes 10x fewer) and is a comparison with the reference, never a "
                            "roofline fraction; bytes_moved_floor = blocks read once + direction written; the timed loop enqueues the steps "
                            "back to back (statistics are fetched once, after it)")
    return out


def config4_leg(build, fence, wd, world, rank, device, use_dist, share_gpu, n=65536, me=0, mi=32768, steps=2, warmup=1):
    """BASELINE.json configs[3] (n=65536, mi=32768 -> KKT dim 131072, 137 GB of KKT storage over the ranks): the size the
    north star's ">= 5x the 1-GPU steps/s at 8 GPUs" is stated on.  1 warm-up + 2 timed steps with the same fences as the
    headline (barrier + synchronize, max over ranks), backward error from the blocks, inertia.  Returned on every rank
    (rank 0 attaches it); {"skipped": why} when the GPUs cannot hold it."""
    import torch
    import torch.distributed as dist
    N = n + 2 * mi + me
    free, total = torch.cuda.mem_get_info(device)
    # peak per rank: the generator's transient (M, M M', Q: 3 n^2) or the resident set (its share of the KKT storage and the
    # panel buffers, its rows of Q / Ji -- all of them on one rank)
    need = 8.0 * max(3.0 * n * n + 2.0 * n * mi, 1.08 * N * N / world + (n * n + n * mi) / world + n * mi) + 8e9
    ok = torch.tensor([1.0 if free >= need else 0.0], dtype=torch.float64, device="cpu" if share_gpu else device)
    if use_dist:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) < 0.5:
        return {"skipped": "needs %.0f GB of free HBM per GPU, %.0f available" % (need / 1e9, free / 1e9), "kkt_dim": N}
    wd.watch("config 4: problem generation + staging", 1200)
    qp, core, one_step = build(n, me, mi, 0)
    wd.watch("config 4: steps", 600 + 300 * (steps + warmup))
    for _ in range(warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    fl = ms = 0.0
    for _ in range(steps):
        dz, st = one_step()
        tm = core.timings()
        fl += tm["trailing_flops"]; ms += tm["trailing_ms"]
    fence()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    raw = dz.clone()
    raw[n + mi:] *= -1.0
    if world > 1:
        g_res = core.residual_dist()
        berr = float((core.matvec_dist(raw) - g_res).norm() / g_res.norm())
    else:
        g_res = core.residual()
        berr = float((core.matvec(raw) - g_res).norm() / g_res.norm())
    out = {"workload": "BASELINE.json configs[3]: synthetic convex dense QP Newton step, n=%d me=%d mi=%d -> KKT dim N=%d, seed 0, nb=%d"
                       % (n, me, mi, N, core.nb if hasattr(core, "nb") else 0),
           "kkt_dim": N, "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
           "value": steps / el, "unit": "steps/s",
           "bulk_update_tflops_rank0": (fl / 1e12) / (ms * 1e-3) if ms > 0 else None,
           "step_flops_dense_equivalent": N ** 3 / 3.0 + 2.0 * N ** 2,
           "trailing_flops_executed_per_step_rank0": fl / steps,
           "backward_error": berr, "inertia": {"n_neg": st["n_neg"], "expected": me + mi, "n_zero": st["n_zero"]},
           "note": "the 1-GPU line's config4.value is the denominator of the north star's >= 5x target, the 8-GPU line's its numerator"}
    core.close()
    del core, qp
    torch.cuda.empty_cache()
    return out


def lbfgs_block(device, n=131072, me=512, mi=1536, m=8, reps=3):
    """One L-BFGS search direction (include/pyipm_lbfgs.h; pyipm.py:1184-1246) on synthetic QP-shaped data, outside the
    timed region of `value`: ms per direction with a fresh Jacobian and with J'J reused, the Gram launch's rate, and the
    size-independent check |H dz - g| / |g| for H = Z - U inv(M) U' applied matrix-free.  Details: tools/bench_lbfgs.py."""
    import torch
    from pyipm_amd.lbfgs import LbfgsCore
    p, N = me + mi, n + 2 * mi + me
    gen = torch.Generator(device=device); gen.manual_seed(1)
    J = torch.randn((n, p), generator=gen, dtype=torch.float64, device=device) / np.sqrt(n)
    S = torch.randn((n, m), generator=gen, dtype=torch.float64, device=device) / np.sqrt(n)
    Mq = torch.randn((n, 8), generator=gen, dtype=torch.float64, device=device) / 3.0
    Y = Mq @ (Mq.t() @ S) + 0.5 * S
    SY = (S.t() @ Y).cpu().numpy()
    SS, L, D = (S.t() @ S).cpu().numpy(), np.tril(SY, -1), np.diag(np.diag(SY))
    zeta = float(SY[-1, -1] / SS[-1, -1])
    u = lambda k, lo, hi: lo + (hi - lo) * torch.rand(k, generator=gen, dtype=torch.float64, device=device)   # noqa: E731
    s, lda = u(mi, 0.5, 2.0), torch.cat([torch.randn(me, generator=gen, dtype=torch.float64, device=device), u(mi, 0.5, 2.0)])
    g = torch.randn(N, generator=gen, dtype=torch.float64, device=device)
    core = LbfgsCore(n, me, mi, m, device=device.index)

    def run(restage):
        ts = []
        for it in range(reps + 1):
            if restage:
                core.stage_jacobian(J[:, :me], J[:, me:])
            dz, st = core.direction(g, s, lda, zeta, S, Y, SS, L, D, reg=1e-12)
            if it:
                ts.append(core.last_timings())
        return dz, st, {k: float(np.median([t[k] for t in ts])) for k in ts[0]}

    dz, st, fresh = run(True)
    _, _, reuse = run(False)
    x, ds, dl = dz[:n], dz[n:n + mi], dz[n + mi:]
    W = torch.cat([zeta * S, Y], dim=1)
    Minv = torch.from_numpy(np.block([[zeta * SS, L], [L.T, -D]])).to(device)
    res = torch.empty_like(dz)
    res[:n] = zeta * x - W @ torch.linalg.solve(Minv, W.t() @ x) + J @ dl
    res[n:n + mi] = lda[me:] / (s + np.finfo(float).eps) * ds - dl[me:]
    low = J.t() @ x
    low[me:] -= ds
    res[n + mi:] = low
    core.close()
    # CPU leg (part of this file's cpu_baseline duty): the reference's arithmetic as restated by oracle/lbfgs_oracle.py on a
    # bounded sample — same p and m, fewer rows: its B' diag(1/A) is a DENSE (n+mi)^2 product (pyipm.py:1102-1104),
    # quadratic in n, so it cannot run at the device size at all
    from oracle import lbfgs_oracle as lo
    nc = 8192
    t0 = time.perf_counter()
    lo.direction(np.concatenate([g[:nc].cpu().numpy(), g[n:].cpu().numpy()]), zeta, S[:nc].cpu().numpy(), Y[:nc].cpu().numpy(),
                 SS, L, D, Je=J[:nc, :me].cpu().numpy() * np.sqrt(n / nc), Ji=J[:nc, me:].cpu().numpy() * np.sqrt(n / nc),
                 s=s.cpu().numpy(), lda=lda.cpu().numpy(), reg=1e-12)
    cpu = {"seconds_per_direction": time.perf_counter() - t0, "n": nc, "me": me, "mi": mi, "m": m, "kind": "port",
           "cores": os.cpu_count(), "sample": "oracle/lbfgs_oracle.py, same p and m, n = %d rows of the %d" % (nc, n)}
    return {"workload": "n=%d me=%d mi=%d m=%d (J = %.1f GB), synthetic" % (n, me, mi, m, n * p * 8 / 1e9), "cpu_baseline": cpu,
            "ms_per_direction": fresh["total_ms"], "ms_per_direction_reusing_gram": reuse["total_ms"],
            "gram_ms": fresh["gram_ms"], "gram_tflops": fresh["gram_flops"] / (fresh["gram_ms"] * 1e-3) / 1e12,
            "factor_ms": fresh["factor_ms"], "jacobian_passes_ms": fresh["jacobian_passes_ms"],
            "residual_H_dz_minus_g_rel": float((res - g).norm() / g.norm()), "regularised": st["regularised"]}


_WD = None      # main()'s watchdog: an exception behind the ladder (the timed region on the rung it chose) still has a number to report


if __name__ == "__main__":
    try:
        main()
    except Exception as e:
        if _WD is not None and _WD.fallback:
            import traceback
            traceback.print_exc()
            _WD.fail_now(_WD.phase or "after the ladder", e)
        raise
