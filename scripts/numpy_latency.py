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
    # the headline batch handed over as NumPy arrays and taken back as NumPy arrays: the PCIe-inclusive rate (never the
    # `value` of bench.py, whose inputs are resident in HBM when its timed region starts)
    n_s, n_u, N, T = 2, 1, 5000, 65536
    prob = workload.make_problem(5, N, n_s, n_u, T)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    l = np.array([0.05, 0.02])
    d = {k: B.as_dev(prob[k], gp.device) for k in ("p", "k_ff", "Q", "k_fb")}
    t_dev = _timeit(lambda: reach.onestep_reachability_batch(d["p"], gp, d["k_ff"], l, l, d["Q"], d["k_fb"], 2.0), n=10, warmup=2)
    t_np = _timeit(lambda: reach.onestep_reachability_batch(prob["p"], gp, prob["k_ff"], l, l, prob["Q"], prob["k_fb"], 2.0), n=10, warmup=2)
    mb = T * 8 * (n_s + n_u + n_s * n_s + n_u * n_s + n_s + n_s * n_s) / 1e6
    print("headline batch (N=5000, T=65536 one-step): device tensors in/out %.2f ms = %.4g evals/s; NumPy in/out %.2f ms = %.4g evals/s "
          "(%.1f MB over PCIe per step)" % (t_dev / 1e3, T / (t_dev * 1e-6), t_np / 1e3, T / (t_np * 1e-6), mb), flush=True)


if __name__ == "__main__":
    main()
