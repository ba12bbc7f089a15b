#!/usr/bin/env python3
"""Trailing-update product alone: TFLOP/s of sr_gemm_tn_upper by size and tile order, plus a correctness check of
both orders against each other and NumPy.  GPU box:  python scripts/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import _buffers as B  # noqa: E402
from safe_exploration_amd._lib import lib, check  # noqa: E402


def run(M, N, K, order, dev, A, Bm, C, reps=3):
    s = B.stream_ptr(dev)
    for _ in range(1):
        check(lib.sr_test_gemm_tn_upper(0, B.ptr(A), A.shape[1], B.ptr(Bm), Bm.shape[1], B.ptr(C), C.shape[1], M, N, K,
                                        -1.0, 1.0, order, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(lib.sr_test_gemm_tn_upper(0, B.ptr(A), A.shape[1], B.ptr(Bm), Bm.shape[1], B.ptr(C), C.shape[1], M, N, K,
                                        -1.0, 1.0, order, s))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tm, tn = M // 128, N // 128
    tiles = tm * tn - tm * (tm - 1) // 2
    return ms, 2.0 * tiles * 128 * 128 * K / ms / 1e9


def main():
    dev = torch.device("cuda", 0)
    # correctness on a ragged shape (partial super-tiles)
    M, N, K = 128 * 11, 128 * 19, 256
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn((K, N), dtype=torch.float64, device=dev, generator=g)
    C0 = torch.randn((M, N), dtype=torch.float64, device=dev, generator=g)
    outs = []
    for order in (0, 1):
        C = C0.clone()
        check(lib.sr_test_gemm_tn_upper(0, B.ptr(A), N, B.ptr(A), N, B.ptr(C), N, M, N, K, -1.0, 1.0, order, B.stream_ptr(dev)))
        outs.append(C)
    ref = C0 - A[:, :M].T @ A
    mask = torch.zeros((M, N), dtype=torch.bool, device=dev)
    for m in range(M // 128):
        mask[m * 128:(m + 1) * 128, m * 128:] = True
    # the 64-tile variant leaves the lower-left quarter of the diagonal blocks alone: compare where both must agree
    for m in range(M // 128):
        mask[m * 128 + 64:(m + 1) * 128, m * 128:m * 128 + 64] = False
    print("order 0 vs ref: %.2e   order 1 vs order 0: %.2e   untouched lower part intact: %s" % (
        float((outs[0] - ref)[mask].abs().max()), float((outs[1] - outs[0]).abs().max()),
        bool((outs[0][~mask] == C0[~mask]).all())), flush=True)
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for (M, N, K) in sizes or ((48000, 48000, 1024), (24000, 24000, 1024), (12800, 12800, 1024), (1024, 48000, 1024),
                      (4096, 4096, 512), (48000, 48000, 512)):
        A = torch.randn((K, N), dtype=torch.float64, device=dev, generator=g)
        C = torch.zeros((M, N), dtype=torch.float64, device=dev)
        for order in (0, 1):
            ms, tf = run(M, N, K, order, dev, A, A, C)
            print("M=%6d N=%6d K=%5d order %d: %9.3f ms  %6.2f TFLOP/s" % (M, N, K, order, ms, tf), flush=True)
        del A, C


if __name__ == "__main__":
    main()
