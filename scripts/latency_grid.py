#!/usr/bin/env python3
"""(N, T) grid of the wall time per predict call (device tensors in/out), pendulum dims; plus one-step and
linearize latencies at the headline model size.  GPU box:  python scripts/latency_grid.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def timeit(fn, n=100):
    return _timeit(fn, n=n, warmup=10)


def main():
    Ns = [int(a) for a in sys.argv[1:]] or [100, 200, 300, 500, 700, 1024, 1500, 2000, 3000, 5000]
    Ts = [1, 4, 16, 32, 64, 128, 256, 512, 1024, 4096]
    print("predict wall time per call [us], pendulum dims (n_out=2, D=3)")
    print("%6s " % "N" + " ".join("%8s" % ("T=%d" % t) for t in Ts))
    for N in Ns:
        prob = workload.make_problem(9, N, 2, 1, max(Ts), sf2=0.01)
        gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        # two passes over the row, the smaller of the two medians per cell: the boxes show disturbances of +5 .. 10 us that
        # last about a second (longer than a cell's five batches; r03 / r04 grids had single rows 25 instead of 14 us)
        xs = [B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device) for T in Ts]
        row = [float("inf")] * len(Ts)
        for _ in range(2):
            for i, x in enumerate(xs):
                row[i] = min(row[i], timeit(lambda: gp.predict_device(x, True)))
        print("%6d " % N + " ".join("%8.0f" % v for v in row), flush=True)
        if N == Ns[-1]:
            l = np.array([0.05, 0.02])
            for T in (1, 32, 128):
                tp, tq, tkff, tkfb = (B.as_dev(prob[k][:T], gp.device) for k in ("p", "Q", "k_ff", "k_fb"))
                print("N=%d T=%d one-step reachability: %.1f us" % (
                    N, T, timeit(lambda: reach.onestep_reachability_batch(tp, gp, tkff, l, l, tq, tkfb, 2.0))))
            x1 = B.as_dev(np.hstack((prob["p"][0], prob["k_ff"][0])), gp.device)
            print("N=%d linearize_device (jacobians=True outputs): %.1f us" % (N, timeit(lambda: gp.linearize_device(x1))))
            print("N=%d __call__ (numpy in/out): %.1f us" % (N, timeit(lambda: gp(prob["p"][:1], prob["k_ff"][:1]))))
        del gp


if __name__ == "__main__":
    main()
