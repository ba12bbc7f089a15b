#!/usr/bin/env python3
"""T = 1 predict on device tensors (asynchronous launches back to back): the streamed one-launch kernel's time by model size;
SR_ST1_PROBE=1 stops the kernel after its streaming part, 2 after the per-column-block reductions (results then invalid:
measurement of where the time goes).  GPU box: python scripts/t1_probe.py [Ns]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: F401  (the switches exist in the lab build only)
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
from _timing import timeit
for N in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000,2000,3000,5000").split(",")]:
    prob = workload.make_problem(9, N, 2, 1, 8, sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    x = B.as_dev(np.hstack((prob["p"][:1], prob["k_ff"][:1])), gp.device)
    t = timeit(lambda: gp.predict_device(x, True), n=400, warmup=30)
    Np = gp._handle.Np
    mb = 2 * Np * (Np + 128) / 2 * 8 / 1e6
    print("SR_ST1_PROBE=%s N=%d: %.1f us per call; upper triangle of U^-1 %.0f MB -> %.2f TB/s" % (os.environ.get("SR_ST1_PROBE", "0"), N, t, mb, mb / t), flush=True)
    del gp
