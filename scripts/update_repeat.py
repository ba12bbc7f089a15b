#!/usr/bin/env python3
"""Wall time of consecutive model updates of one model (first call allocates): python scripts/update_repeat.py N reps [panel]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402

N = int(sys.argv[1])
reps = int(sys.argv[2])
panel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prob = workload.make_problem(4, N, 2, 1, 16)
gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
gp.set_fact_panel(panel)
out = []
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t0)
fl = 2 * (2.0 / 3.0) * float(N) ** 3
print("N=%d panel=%d" % (N, panel), " ".join("%.3fs(%.1fTF)" % (t, fl / t / 1e12) for t in out), flush=True)
