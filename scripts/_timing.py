"""Timing helper of the latency scripts: the MEDIAN of several batches, not one long mean.

Round 3's tables carried single-batch means; three of their entries were two to three times their neighbours (a
linearize call at n_s = 4, N = 500: 68 us beside 32 us; two one-step entries) and did not reproduce: one stall inside a
200-call batch (an allocation, a clock ramp, a host hiccup) moves a mean by tens of percent and a median not at all."""
import statistics
import time

import torch


def timeit(fn, n=200, warmup=10, batches=5, sync=True):
    """us per call: `batches` batches of n / batches calls each, synchronised at both ends; the median batch."""
    for _ in range(warmup):
        fn()
    per = max(1, n // batches)
    res = []
    for _ in range(batches):
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        if sync:
            torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / per * 1e6)
    return statistics.median(res)
