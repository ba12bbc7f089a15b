#!/usr/bin/env python3
"""Wall time per linearize_device call (mu, var, d mu/dx, d var/dx, Hessian of mu; device tensors) by model size.
GPU box:  python scripts/linearize_bench.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def timeit(fn, n=200):
    return _timeit(fn, n=n, warmup=10)


def main():
    args = list(sys.argv[1:])
    kt = "rbf"
    if "--kern" in args:                                    # rbf (default) | mat52 | lin_rbf | lin_mat52
        i = args.index("--kern")
        kt = args[i + 1]
        del args[i:i + 2]
    Ns = [int(a) for a in args] or ([200, 500, 700, 1024, 1500, 2000, 3000, 5000] if kt == "rbf" else [25, 150, 350, 500, 1000, 5000])
    print("# linearize_device (sr_gp_linearize on device tensors, asynchronous launches back to back), kern_types = ['%s'] * n_out" % kt)
    for n_s, n_u in ((2, 1), (4, 1)):
        row = []
        for N in Ns:
            prob = workload.make_problem(9, N, n_s, n_u, 4, sf2=0.01)
            if kt == "rbf":
                hyp = workload.hyp_list(prob)
            else:
                from call_latency import kern_hyp
                hyp = kern_hyp(kt, np.random.default_rng(N), n_s + n_u, n_s)
            gp = SimpleGPModel(n_s, n_s, n_u, kern_types=[kt] * n_s, hyp=hyp, device="cuda:0")
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            x1 = B.as_dev(np.hstack((prob["p"][0], prob["k_ff"][0])), gp.device)
            row.append(timeit(lambda: gp.linearize_device(x1)))
            del gp
        print("n_s=%d n_u=%d  " % (n_s, n_u) + "  ".join("N=%d: %.1f us" % (N, v) for N, v in zip(Ns, row)), flush=True)


if __name__ == "__main__":
    main()
