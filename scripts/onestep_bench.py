#!/usr/bin/env python3
"""Wall time per one-step reachability call (device tensors) of small models: posterior + ellipsoid step as two
launches against the one-launch route (persistent chain kernel with H = 1).  GPU box:  python scripts/onestep_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def timeit(fn, n=200):
    return _timeit(fn, n=n, warmup=10)


def main():
    print("%5s %4s %6s %14s %14s" % ("N", "n_s", "T", "two launches us", "one launch us"))
    for n_s, N in ((2, 100), (2, 200), (4, 150), (2, 350), (2, 500)):
        Tmax = 2048
        prob = workload.make_problem(9, N, n_s, 1, Tmax, sf2=0.01)
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        l = np.full(n_s, 0.05)
        for T in (1, 16, 256, 480, 960, 1920):
            tp, tq, tkff, tkfb = (B.as_dev(prob[k][:T], gp.device) for k in ("p", "Q", "k_ff", "k_fb"))
            fn = lambda: reach.onestep_reachability_batch(tp, gp, tkff, l, l, tq, tkfb, 2.0)
            out = []
            for on in (False, True):
                gp.set_chain(on)
                out.append(timeit(fn))
                took = gp.last_chain
            print("%5d %4d %6d %14.1f %14.1f%s" % (N, n_s, T, out[0], out[1], "" if took else "  (not taken)"), flush=True)
        del gp


if __name__ == "__main__":
    main()
