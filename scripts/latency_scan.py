#!/usr/bin/env python3
"""predict wall time over a FINE grid of model sizes (device tensors, launches back to back): a planner that picks an unlucky
launch shape at some size shows as a bump in a column.  GPU box: python scripts/latency_scan.py [N0 N1 step] [Ts]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if any(k.startswith(("SR_ST_", "SR_BAL_")) for k in os.environ):
    import _lab  # noqa: F401  (the switches exist in the lab build only)
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
from _timing import timeit
n0, n1, st = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (600, 10000, 200)))
Ts = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "1,4,16,32,64,128,256,1024").split(",")]
print("%6s " % "N" + " ".join("%8s" % ("T=%d" % T) for T in Ts))
for N in range(n0, n1 + 1, st):
    prob = workload.make_problem(9, N, 2, 1, max(Ts), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    row = []
    for T in Ts:
        x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
        row.append(timeit(lambda: gp.predict_device(x, True), n=120, warmup=10))
    print("%6d " % N + " ".join("%8.1f" % v for v in row), flush=True)
    del gp
