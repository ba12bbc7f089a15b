#!/usr/bin/env python3
"""Refit wall time by model size in one process (median of 9 updates); environment knobs of the library (SR_T64_*, SR_FACT_*)
are read at the first call, so one process per setting.  GPU box: SR_T64_JOBS_THR=512 python scripts/refit_ab.py 2000,5000,10000"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: F401  (the switches exist in the lab build only)
from safe_exploration_amd import SimpleGPModel, workload
out = []
for N in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000,2000,5000,10000").split(",")]:
    prob = workload.make_problem(4, N, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    for _ in range(3):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    ts = []
    for _ in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    out.append("N=%d %.3f ms" % (N, statistics.median(ts)))
    del gp
print(" ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("SR_")) or "defaults", "|", "  ".join(out), flush=True)
