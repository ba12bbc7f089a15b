#!/usr/bin/env python3
"""A model whose size changes from call to call (the exploration loop adds points after every episode): wall time of
train() at a new N (a new handle: device blocks from the library's cache, process-wide update streams) and of
update_model(replace_old=False) crossing padded sizes.  GPU box:  python scripts/growing_model.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    prob = workload.make_problem(4, N + 1100, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    Z, Y = prob["Z"], prob["Y"]

    def timed(f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for n in (N, N, N + 130, N + 130, N + 260, N, N + 390, N + 520, N + 650, N + 780):
        print("train at N = %5d: %8.2f ms" % (n, timed(lambda: gp.train(Z[:n], Y[:n], opt_hyp=False))), flush=True)
    gp.train(Z[:N], Y[:N], opt_hyp=False)
    n = N
    for m in (100,) * 10:
        t = timed(lambda: gp.update_model(Z[n:n + m], Y[n:n + m], opt_hyp=False, replace_old=False))
        n += m
        print("update_model +%d -> N = %5d: %6.2f ms" % (m, n, t), flush=True)


if __name__ == "__main__":
    main()
