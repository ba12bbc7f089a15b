#!/usr/bin/env python3
"""Latency of the small-batch entry points (the regime the CasADi/IPOPT loop drives): wall time per
call of predict / onestep / linearize for T in {1, 16, 128, 1024} at N in {200, 5000}."""
import json
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
    out = {}
    for N in (200, 5000):
        prob = workload.make_problem(9, N, 2, 1, 1024, sf2=0.01)
        gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        dev = gp.device
        l = np.array([0.05, 0.02])
        for T in (1, 16, 128, 1024):
            x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), dev)
            tp, tq, tkff, tkfb = (B.as_dev(prob[k][:T], dev) for k in ("p", "Q", "k_ff", "k_fb"))
            out["N%d_T%d_predict_us" % (N, T)] = round(timeit(lambda: gp.predict_device(x, True)), 1)
            out["N%d_T%d_onestep_us" % (N, T)] = round(
                timeit(lambda: reach.onestep_reachability_batch(tp, gp, tkff, l, l, tq, tkfb, 2.0)), 1)
        # HBM roofline of the single-query path: bytes of U^-1 streamed / time of the variance launches
        x = B.as_dev(np.hstack((prob["p"][:1], prob["k_ff"][:1])), dev)
        gp.prof_reset(); gp.prof_enable(True)
        for _ in range(20):
            gp.predict_device(x, True)
        gp.prof_enable(False)
        from safe_exploration_amd import _lib
        ms, n = gp.prof_get(_lib.K_VAR)
        Np = gp._handle.Np
        if n == 0:                       # small model: the one-launch pass (sr_small.hip) ran instead
            ms, n = gp.prof_get(_lib.K_SMALL)
            out["N%d_T1_fused_kernel_us" % N] = round(1e3 * ms / max(n, 1), 1)
        else:
            out["N%d_T1_var_kernels_us" % N] = round(1e3 * ms / max(n, 1), 1)
            out["N%d_T1_var_GBps" % N] = round(2 * (Np * (Np + 1) / 2) * 8 / (ms / max(n, 1) * 1e-3) / 1e9, 1)
        x1 = B.as_dev(np.hstack((prob["p"][0], prob["k_ff"][0])), dev)
        out["N%d_linearize_us" % N] = round(timeit(lambda: gp.linearize_device(x1)), 1)
        out["N%d_call_numpy_us" % N] = round(timeit(lambda: gp(prob["p"][:1], prob["k_ff"][:1])), 1)
        roll = workload.random_rollout_controls(3, 256, 15, 2, 1)
        tr = {k: B.as_dev(v, dev) for k, v in roll.items()}
        a = 0.8 * np.eye(2)
        out["N%d_multistep_T256_H15_us" % N] = round(timeit(
            lambda: reach.multistep_reachability_batch(tr["p0"], gp, tr["k_fb"], tr["k_ff"], l, l, None, 2.0, a,
                                                       np.zeros((2, 1))), 20), 1)
    # model update: block row append vs refactorisation (exploration adds a few samples per episode)
    prob = workload.make_problem(13, 5050, 2, 1, 8, sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    t0 = time.perf_counter(); gp.train(prob["Z"][:5000], prob["Y"][:5000], opt_hyp=False); torch.cuda.synchronize()
    out["N5000_first_fit_ms"] = round(1e3 * (time.perf_counter() - t0), 2)      # allocates factors, scratch, streams
    t0 = time.perf_counter(); gp.train(prob["Z"][:5000], prob["Y"][:5000], opt_hyp=False); torch.cuda.synchronize()
    out["N5000_refit_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    t0 = time.perf_counter()
    gp.update_model(prob["Z"][5000:], prob["Y"][5000:], opt_hyp=False, replace_old=False)
    torch.cuda.synchronize()
    out["N5000_append50_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
