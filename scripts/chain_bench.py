#!/usr/bin/env python3
"""Wall time of multi-step reachability chains of small models: persistent kernel against per-step launches.
GPU box:  python scripts/chain_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def timeit(fn, n=50):
    return _timeit(fn, n=n, warmup=5)


def main():
    print("%5s %4s %6s %4s %12s %12s" % ("N", "n_s", "T", "H", "per-step us", "one launch us"))
    for n_s, n_u, N, T, H in ((2, 1, 200, 256, 15), (2, 1, 200, 16, 15), (2, 1, 200, 1024, 15), (2, 1, 100, 256, 15),
                              (2, 1, 350, 256, 15), (2, 1, 500, 256, 15), (4, 1, 150, 256, 15), (4, 1, 150, 960, 15),
                              (2, 1, 200, 256, 5), (2, 1, 200, 768, 15), (2, 1, 200, 1536, 15), (2, 1, 200, 1920, 15),
                              (2, 1, 200, 2304, 15), (2, 1, 200, 3072, 15), (2, 1, 200, 4096, 15), (4, 1, 150, 416, 15),
                              (4, 1, 150, 832, 15), (4, 1, 150, 1248, 15), (2, 1, 500, 416, 15), (2, 1, 500, 832, 15)):
        prob = workload.make_problem(9, N, n_s, n_u, T, sf2=0.01)
        gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        rng = np.random.default_rng(0)
        l = np.full(n_s, 0.05)
        k_ff = B.as_dev(0.1 * rng.standard_normal((T, H, n_u)), gp.device)
        k_fb = B.as_dev(0.1 * rng.standard_normal((T, H - 1, n_u, n_s)), gp.device)
        p0 = B.as_dev(prob["p"][:T], gp.device)
        fn = lambda: reach.multistep_reachability_batch(p0, gp, k_fb, k_ff, l, l, None, 2.0)
        out = []
        for on in (False, True):
            gp.set_chain(on)
            out.append(timeit(fn))
        print("%5d %4d %6d %4d %12.1f %12.1f%s" % (N, n_s, T, H, out[0], out[1], "" if gp.last_chain else "   (per-step launches)"),
              flush=True)
        del gp


if __name__ == "__main__":
    main()
