#!/usr/bin/env python3
"""Wall time of the blocking single-query host entry points (NumPy in / out: what a CasADi callback pays).
GPU box:  python scripts/call_latency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B  # noqa: E402


def t(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    for n_s, N in ((2, 100), (2, 200), (4, 150), (2, 500), (2, 2000), (2, 5000)):
        prob = workload.make_problem(9, N, n_s, 1, 4, sf2=0.01)
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        io = gp._handle.single_io()
        x = B.as_dev(np.hstack((prob["p"][:1], prob["k_ff"][:1])), gp.device)
        row = {}
        for mode, (mb, di) in (("sync", (False, False)), ("mailbox", (True, False)), ("direct", (True, True))):
            io["mailbox"], io["direct"] = mb, di
            row[mode] = (t(lambda: gp(prob["p"][:1], prob["k_ff"][:1])),
                         t(lambda: gp.linearize_predict(prob["p"][:1], prob["k_ff"][:1], True)))
            if mode == "direct" and not io["direct"]:
                row[mode] = (float("nan"), float("nan"))
        print("n_s=%d N=%5d  __call__: copy+sync %.1f us, mailbox %.1f, one command %.1f | linearize_predict(jacobians=True): "
              "%.1f, %.1f, %.1f us | kernel alone (async) %.1f us" % (
                  n_s, N, row["sync"][0], row["mailbox"][0], row["direct"][0], row["sync"][1], row["mailbox"][1],
                  row["direct"][1], t(lambda: gp.predict_device(x, True))), flush=True)
        del gp


if __name__ == "__main__":
    main()
