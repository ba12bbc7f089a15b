#!/usr/bin/env python3
"""A/B of the few-query-tile routes of the variance contraction: predict wall time per call with the default route (balanced shares K2b) against the
split-K chunks K2k (set_small_path(1 + 4)), and the largest difference between their variances (the XCD slabs K2x of round 4 are gone).
GPU box: python scripts/splitk_ab.py [Ns] [Ts]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3700,4000,4500,5000").split(",")]
Ts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128,256,384").split(",")]
print("%6s %5s %10s %10s %12s" % ("N", "T", "default", "chunks K2k", "max |dvar|"))
for N in Ns:
    prob = workload.make_problem(9, N, 2, 1, max(256, max(Ts)), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    for T in Ts:
        x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
        res, var = [], []
        for mode in (1, 5):
            gp.set_small_path(mode)
            for _ in range(10):
                out = gp.predict_device(x, True)
            torch.cuda.synchronize()
            var.append(out[1].clone())
            t0 = time.perf_counter()
            for _ in range(200):
                gp.predict_device(x, True)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 200 * 1e6)
        print("%6d %5d %10.1f %10.1f %12.3e" % (N, T, res[0], res[1], float((var[0] - var[1]).abs().max())), flush=True)
    del gp
