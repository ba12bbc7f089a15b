#!/usr/bin/env python3
"""Row-append latency: update_model(replace_old=False) with m new points on an N-point model vs a refit.
GPU box:  python scripts/append_bench.py [N]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    out = {}
    for m in (1, 16, 17, 50, 96, 128):
        prob = workload.make_problem(13, N + 3 * m, 2, 1, 8, sf2=0.01)
        gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"][:N], prob["Y"][:N], opt_hyp=False)
        torch.cuda.synchronize()
        ts = []
        for r in range(3):
            lo = N + r * m
            t0 = time.perf_counter()
            gp.update_model(prob["Z"][lo:lo + m], prob["Y"][lo:lo + m], opt_hyp=False, replace_old=False)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        # against a refit on the same data
        ref = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
        ref.train(prob["Z"][:N + 3 * m], prob["Y"][:N + 3 * m], opt_hyp=False)
        x = np.hstack((prob["p"], prob["k_ff"]))
        a, b = gp.predict(x), ref.predict(x)
        out["N%d_append%d_ms" % (N, m)] = [round(t, 3) for t in ts]
        out["N%d_append%d_err" % (N, m)] = [float(np.abs(a[0] - b[0]).max()), float(np.abs(a[1] - b[1]).max())]
        del gp, ref
    print(json.dumps(out))


if __name__ == "__main__":
    main()
