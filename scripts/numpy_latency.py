#!/usr/bin/env python3
"""Wall time of the entry points when they are called the way the reference's callers call them: NumPy arrays in, NumPy
arrays out (one pinned block in, one packed block back: _buffers.Staging), beside the same call on device tensors.
GPU box:  python scripts/numpy_latency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def timeit(fn, n=300):
    return _timeit(fn, n=n, warmup=20)


def main():
    print("%4s %6s %5s | %28s | %28s | %28s" % ("n_s", "N", "T", "predict  dev / NumPy [us]", "one-step  dev / NumPy / ref", "15-step chain  dev / NumPy / ref"))
    for n_s, N, T in ((2, 200, 1), (2, 200, 256), (4, 150, 1), (4, 150, 256), (2, 5000, 1), (2, 5000, 16)):
        n_u, H = 1, 15
        prob = workload.make_problem(9, N, n_s, n_u, max(T, 8), sf2=0.01)
        gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        rng = np.random.default_rng(0)
        l = np.full(n_s, 0.05)
        p, kff, Q, kfb = prob["p"][:T], prob["k_ff"][:T], prob["Q"][:T], prob["k_fb"][:T]
        kff_h = 0.1 * rng.standard_normal((T, H, n_u))
        kfb_h = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
        x = np.hstack((p, kff))
        d = {k: B.as_dev(v, gp.device) for k, v in dict(p=p, kff=kff, Q=Q, kfb=kfb, kff_h=kff_h, kfb_h=kfb_h, x=x).items()}
        pr = (timeit(lambda: gp.predict_device(d["x"], True)), timeit(lambda: gp.predict(x, None, True)))
        one = [timeit(lambda: reach.onestep_reachability_batch(d["p"], gp, d["kff"], l, l, d["Q"], d["kfb"], 2.0)),
               timeit(lambda: reach.onestep_reachability_batch(p, gp, kff, l, l, Q, kfb, 2.0)), float("nan")]
        ms = [float("nan")] * 3
        if N <= 512:
            ms[0] = timeit(lambda: reach.multistep_reachability_batch(d["p"], gp, d["kfb_h"], d["kff_h"], l, l, None, 2.0))
            ms[1] = timeit(lambda: reach.multistep_reachability_batch(p, gp, kfb_h, kff_h, l, l, None, 2.0))
        if T == 1:
            one[2] = timeit(lambda: reach.onestep_reachability(p[0].reshape(-1, 1), gp, kff[0].reshape(-1, 1), l, l, Q[0],
                                                               kfb[0], 2.0, 0))
            if N <= 512:
                ms[2] = timeit(lambda: reach.multistep_reachability(p[0].reshape(-1, 1), gp, kfb_h[0], kff_h[0], l, l,
                                                                    None, 2.0, 0))
        print("%4d %6d %5d | %13.1f %14.1f | %8.1f %9.1f %9.1f | %8.1f %9.1f %9.1f" % (
            (n_s, N, T) + pr + tuple(one) + tuple(ms)), flush=True)
        del gp


if __name__ == "__main__":
    main()
