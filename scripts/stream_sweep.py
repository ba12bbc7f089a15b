#!/usr/bin/env python3
"""predict wall time per call for a few (N, T) of the streamed small-batch route.  GPU box: python scripts/stream_sweep.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024,2000,3000,5000").split(",")]
Ts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8,16,32,64,96,128").split(",")]
print("KC=%s MAX_T=%s" % (os.environ.get("SR_STREAM_KC"), os.environ.get("SR_STREAM_MAX_T")))
print("%6s" % "N" + "".join("%8s" % ("T=%d" % t) for t in Ts))
for N in Ns:
    prob = workload.make_problem(9, N, 2, 1, max(256, max(Ts)), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    row = []
    for T in Ts:
        x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
        for _ in range(10):
            gp.predict_device(x, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            gp.predict_device(x, True)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 200 * 1e6)
    print("%6d" % N + "".join("%8.1f" % v for v in row), flush=True)
    del gp
