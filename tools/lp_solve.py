#!/usr/bin/env python
"""A dense LP through the device-resident interior-point loop (QPDeviceIPM with Q = 0): min c'x over the box [-1, 1]^n cut by random
half-spaces.  Every Newton system has a zero x-x block (static pivots + refinement on every iterate).  Prints one JSON line; the
optimum is checked against scipy's HiGHS, whose wall time on the host cores is reported beside it (a different algorithm on
different hardware: context, not a baseline).  usage: python tools/lp_solve.py [n] [cuts]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.optimize import linprog

from pyipm_amd.qp import QPDeviceIPM

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(6)
R = rng.standard_normal((extra, n)) / np.sqrt(n)
G = np.vstack([np.eye(n), -np.eye(n), R])
h = np.concatenate([-np.ones(2 * n), -rng.uniform(0.5, 1.5, extra)])
c = rng.standard_normal(n)
p = QPDeviceIPM(np.zeros((n, n)), c, G=G, h=h, verbosity=-1, Ktol=1e-7)
torch.cuda.synchronize()
t0 = time.perf_counter()
x, s, lda, fval, kkt = p.solve()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t0 = time.perf_counter()
ref = linprog(c, A_ub=-G, b_ub=-h, bounds=[(None, None)] * n, method="highs")
dt_ref = time.perf_counter() - t0
b = p.backend
print(json.dumps({
    "workload": "dense LP, n=%d, %d box + %d cut inequalities -> KKT dim %d" % (n, 2 * n, extra, b.core.N),
    "signal": p.signal, "iterations": p.iter_count, "factorisations": b.n_factor, "statically_pivoted_factors": b.n_static,
    "solve_seconds": dt, "newton_seconds": p.timings["newton_s"], "search_seconds": p.timings["search_s"],
    "objective": float(fval), "objective_highs": float(ref.fun), "abs_difference": abs(float(fval) - float(ref.fun)),
    "min_slack": float((G @ x.cpu().numpy() - h).min()),
    "highs_seconds_on_host": dt_ref, "host_cores": os.cpu_count()}))
