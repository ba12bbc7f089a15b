#!/usr/bin/env python3
"""Wall time per `predict` call (device tensors in and out, mean + variance + mean-Jacobian) over a grid of model
sizes N and batch sizes T: which dispatch path (K0, K2s, K2k, K2) serves which cell is decided in gp_pass.
Run on the GPU box:  python scripts/latency_grid.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B  # noqa: E402

TS = (1, 16, 32, 128, 256, 512, 1024, 4096)
print("predict wall time per call [us], pendulum dims (n_out=2, D=3)")
print("%6s " % "N" + " ".join("%8s" % ("T=%d" % t) for t in TS))
for N in (100, 200, 300, 500, 700, 1024, 1500, 2000, 3000, 5000):
    prob = workload.make_problem(9, N, 2, 1, max(TS), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    row = []
    for T in TS:
        x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
        for _ in range(5):
            gp.predict_device(x, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            gp.predict_device(x, True)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 50 * 1e6)
    print("%6d " % N + " ".join("%8.0f" % v for v in row), flush=True)
