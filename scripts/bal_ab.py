#!/usr/bin/env python3
"""predict wall time of the few-query-tile regime (balanced shares K2b) by model size and batch; the number of
workgroups comes from the environment (SR_BAL_WGS / SR_BAL_THR) so that a shell loop can A/B it.
GPU box:  SR_BAL_WGS=512 python scripts/bal_ab.py 5000 128,256,512,1024"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: F401  (the switches exist in the lab build only)
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
from _timing import timeit
Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5000").split(",")]
Ts = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128,256,512,1024").split(",")]
for N in Ns:
    prob = workload.make_problem(9, N, 2, 1, max(Ts), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    row = []
    for T in Ts:
        x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
        row.append(timeit(lambda: gp.predict_device(x, True), n=300, warmup=20))
    print("SR_BAL_WGS=%s SR_BAL_THR=%s N=%d  " % (os.environ.get("SR_BAL_WGS", "-"), os.environ.get("SR_BAL_THR", "-"), N)
          + "  ".join("T=%d: %.1f us" % (T, v) for T, v in zip(Ts, row)), flush=True)
    del gp
