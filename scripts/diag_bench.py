#!/usr/bin/env python3
"""Time the diagonal-block kernel of the factorisation alone (200 back-to-back launches on one input) and check factor
and inverse against NumPy.  GPU box:  python scripts/diag_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _buffers as B  # noqa: E402
from safe_exploration_amd._lib import lib, check  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    M = rng.standard_normal((128, 160))
    A = M.dot(M.T) / 160 + 0.5 * np.eye(128)
    R = np.linalg.cholesky(A).T
    s = B.stream_ptr(dev)
    for skip, name in ((0, "factor + inverse in one sweep"),):
        tA = B.as_dev(A.copy(), dev)
        wt, w = B.empty((128, 128), dev).fill_(7.0), B.empty((128, 128), dev).fill_(7.0)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.sr_test_potrf_diag(0, B.ptr(tA), 128, B.ptr(wt), B.ptr(w), 128, B.ptr(info), skip, s))
        U = B.to_numpy(tA)
        print("%-42s max|U - chol| = %.3e   max|U^-1 - inv| = %.3e   max|w - wt^T| = %.3e  info=%d" % (
            name, np.abs(U - R).max(), np.abs(B.to_numpy(wt) - np.linalg.inv(R)).max(),
            np.abs(B.to_numpy(w) - B.to_numpy(wt).T).max(), int(info.item())))
    for skip, name in ((64, "input untouched"), (0, "refactors its own output")):
        tA = B.as_dev(A.copy(), dev)
        wt, w = B.empty((128, 128), dev), B.empty((128, 128), dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        for _ in range(5):
            lib.sr_test_potrf_diag(0, B.ptr(tA), 128, B.ptr(wt), B.ptr(w), 128, B.ptr(info), skip, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        for _ in range(n):
            lib.sr_test_potrf_diag(0, B.ptr(tA), 128, B.ptr(wt), B.ptr(w), 128, B.ptr(info), skip, s)
        e1.record()
        torch.cuda.synchronize()
        print("skip=%3d  %-40s %7.2f us / launch   (info=%d)" % (skip, name, 1e3 * e0.elapsed_time(e1) / n, int(info.item())))


if __name__ == "__main__":
    main()
