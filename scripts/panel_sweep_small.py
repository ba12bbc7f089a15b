#!/usr/bin/env python3
"""Model update by Cholesky panel width at the chain-bound sizes, in one process (median of 7 updates per setting).
GPU box: python scripts/panel_sweep_small.py [sizes] [panels]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000,2000,3500,5000,7500,10000").split(",")]
panels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,6,8").split(",")]
for N in sizes:
    prob = workload.make_problem(4, N, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    for _ in range(3):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    row = []
    for p in panels:
        gp.set_fact_panel(p)
        ts = []
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        row.append("%d: %.3f" % (p, statistics.median(ts[1:])))
    print("N=%-6d nb=%-3d  ms by panel width (0 = table)  %s" % (N, (N + 2 + 127) // 128, "   ".join(row)), flush=True)
    del gp
