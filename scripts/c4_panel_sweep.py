#!/usr/bin/env python3
"""Model update at N = 50000 by Cholesky panel width, IN ONE PROCESS (boxes differ by 5 % on this workload: only same-box
numbers compare).  GPU box: python scripts/c4_panel_sweep.py [N] [panels]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
panels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,24,32,48,64,96").split(",")]
prob = workload.make_problem(4, N, 2, 1, 16)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
gp.train(prob["Z"], prob["Y"], opt_hyp=False)
gp.train(prob["Z"], prob["Y"], opt_hyp=False)
flops = 2 * (2.0 / 3.0) * float(N) ** 3
for rnd in range(2):
    for p in panels:
        gp.set_fact_panel(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("round %d  panel %3d (0 = by size): %.3f s  %.2f TF" % (rnd, p, dt, flops / dt / 1e12), flush=True)
