#!/usr/bin/env python3
"""Wall time of the blocking single-query host entry points (NumPy in / out: what a CasADi callback pays).
GPU box:  python scripts/call_latency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, workload, _buffers as B  # noqa: E402


from _timing import timeit as _timeit  # noqa: E402  (median of batches)


def t(fn, n=300):
    return _timeit(fn, n=n, warmup=20)


SIZES = ((2, 25), (2, 100), (4, 150), (2, 200), (2, 350), (2, 500), (2, 1000), (2, 2000), (2, 5000))
# --kern mat52 | lin_rbf | lin_mat52: the journal experiments' kernels (defaultconfig_episode.py:39: lin_mat52, m = 150,
# 4 outputs, GP inputs = 3 transformed states + 1 action -> (n_s_out, n_s_in, n_u) = (4, 3, 1), D = 4; untransformed: D = 5)
JOURNAL_SIZES = ((4, 3, 1, 25), (4, 4, 1, 25), (4, 3, 1, 150), (4, 4, 1, 150), (2, 2, 1, 100), (2, 2, 1, 350), (2, 2, 1, 500),
                 (4, 3, 1, 1000))


def kern_hyp(kt, rng, D, n_out):
    """hyper-parameters with the reference's key names (ssm_gpy/gaussian_process.py:491-544)"""
    out = []
    for _ in range(n_out):
        if kt in ("rbf", "mat52"):
            out.append({"lengthscale": rng.uniform(0.5, 1.5, D), "variance": float(rng.uniform(0.5, 1.5)), "noise_variance": 0.02})
        else:
            st = "rbf" if kt == "lin_rbf" else "mat52"
            out.append({"prod.%s.lengthscale" % st: np.array([rng.uniform(0.5, 1.5)]), "prod.%s.variance" % st: float(rng.uniform(0.5, 1.5)),
                        "prod.linear.variances": np.array([rng.uniform(0.5, 1.5)]), "linear.variances": rng.uniform(0.2, 1.0, D),
                        "noise_variance": 0.02})
    return out


def cpu_time_k(Z, Y, kt, hyp, x, second_order, reps=200):
    """as cpu_time, for the general kernels (the oracle's closed forms of the same call)"""
    from oracle import oracle_np as orc
    n = Y.shape[1]
    hy = [{k: v for k, v in h.items() if k != "noise_variance"} for h in hyp]
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * n, hy, np.full(n, 0.02 + 1e-5))
    if second_order:
        fn = lambda: (orc.gp_predict_k(x[None], Z, beta, inv_K, [kt] * n, hy), orc.gp_mean_jacobian_k(x[None], Z, beta, [kt] * n, hy),
                      orc.gp_linearize_extras_k(x, Z, beta, inv_K, [kt] * n, hy))
    else:
        fn = lambda: (orc.gp_predict_k(x[None], Z, beta, inv_K, [kt] * n, hy), orc.gp_mean_jacobian_k(x[None], Z, beta, [kt] * n, hy))
    for _ in range(3):
        fn()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


def main_kern(kt):
    print("# blocking single-query entry points with kern_types = ['%s'] * n_out (the journal experiments' kernel family), wall time "
          "per call [us]; NumPy in / out.  Routes as in the ARD-RBF table: copy + sync, mailbox, one command (sr_gp_call1), RESIDENT "
          "SERVER; CPU: the oracle's NumPy closed forms of the same call (1 BLAS thread / all %d cores; model fit excluded)."
          % (kt, os.cpu_count() or 1))
    from threadpoolctl import threadpool_limits
    cpu_rows = []
    for n_out, n_in, n_u, N in JOURNAL_SIZES:
        rng = np.random.default_rng(100 + N + n_in)
        D = n_in + n_u
        Z = rng.uniform(-1, 1, (N, D))
        Y = rng.standard_normal((N, n_out))
        hyp = kern_hyp(kt, rng, D, n_out)
        gp = SimpleGPModel(n_out, n_in, n_u, kern_types=[kt] * n_out, hyp=hyp, device="cuda:0")
        gp.train(Z, Y, opt_hyp=False)
        io = gp._handle.single_io()
        xq = rng.uniform(-0.6, 0.6, (1, D))
        st, ac = xq[:, :n_in], xq[:, n_in:]
        row = {}
        for mode, (mb, di) in (("sync", (False, False)), ("mailbox", (True, False)), ("direct", (True, True))):
            io["mailbox"], io["direct"] = mb, di
            io.pop("direct_off", None)
            row[mode] = (t(lambda: gp(st, ac)), t(lambda: gp.linearize_predict(st, ac, True)))
            if mode == "direct" and not io["direct"]:
                row[mode] = (float("nan"), float("nan"))
        row["server"] = (float("nan"), float("nan"))
        if gp.start_server(idle_timeout_s=0.05):
            row["server"] = (t_host(lambda: gp(st, ac)), t_host(lambda: gp.linearize_predict(st, ac, True)))
            gp.stop_server()
        print("%s n_out=%d D=%d N=%5d  __call__: copy+sync %.1f, mailbox %.1f, one command %.1f, resident server %.1f || "
              "linearize_predict(jacobians=True): %.1f, %.1f, %.1f, server %.1f" % (
                  kt, n_out, D, N, row["sync"][0], row["mailbox"][0], row["direct"][0], row["server"][0], row["sync"][1],
                  row["mailbox"][1], row["direct"][1], row["server"][1]), flush=True)
        cpu_rows.append((n_out, D, N, Z, Y, hyp, xq[0]))
        del gp
    for n_out, D, N, Z, Y, hyp, x in cpu_rows:
        reps = 100 if N <= 500 else 20
        with threadpool_limits(limits=1):
            one = (cpu_time_k(Z, Y, kt, hyp, x, False, reps), cpu_time_k(Z, Y, kt, hyp, x, True, reps))
        allc = (cpu_time_k(Z, Y, kt, hyp, x, False, reps), cpu_time_k(Z, Y, kt, hyp, x, True, reps))
        print("%s n_out=%d D=%d N=%5d  CPU NumPy (oracle): __call__ %.1f us with 1 BLAS thread, %.1f with all cores; "
              "linearize_predict(jacobians=True) %.1f / %.1f us" % (kt, n_out, D, N, one[0], allc[0], one[1], allc[1]), flush=True)


def cpu_time(prob, second_order, reps=200):
    """The oracle's NumPy evaluation of the same single query (explicit inv_K route of the reference, BLAS threads as the
    box gives them): what a CPU pays for the call the GPU columns time.  Model fit excluded."""
    from oracle import oracle_np as orc
    beta, inv_K, _ = orc.gp_fit(prob["Z"], prob["Y"], prob["lengthscale"], prob["signal_var"], prob["noise_var"] + 1e-5)
    x = np.hstack((prob["p"][0], prob["k_ff"][0]))
    if second_order:
        fn = lambda: (orc.gp_predict(x[None], prob["Z"], beta, inv_K, prob["lengthscale"], prob["signal_var"], True),
                      orc.gp_linearize_extras(x, prob["Z"], beta, inv_K, prob["lengthscale"], prob["signal_var"]))
    else:
        fn = lambda: orc.gp_predict(x[None], prob["Z"], beta, inv_K, prob["lengthscale"], prob["signal_var"], True)
    for _ in range(5):
        fn()
    best = float("inf")
    for _ in range(3):                         # (the best of three batches: the first touches of a thread pool show otherwise)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


def t_host(fn, n=300):
    return _timeit(fn, n=n, warmup=20, sync=False)


def main():
    print("# blocking single-query entry points, wall time per call [us]; NumPy in / out.  GPU routes: copy + sync (H2D copy, "
          "kernel, D2H copy, stream sync), mailbox (results published into pinned memory by a kernel), one command (query in the "
          "kernel arguments, results written to pinned memory by the posterior kernel), RESIDENT SERVER (no launch: mailbox "
          "polled by resident workgroups).  CPU: the oracle's NumPy evaluation of the same call on this box's host cores "
          "(%d; model fit excluded)." % (os.cpu_count() or 1))
    for n_s, N in SIZES:
        prob = workload.make_problem(9, N, n_s, 1, 4, sf2=0.01)
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        io = gp._handle.single_io()
        x = B.as_dev(np.hstack((prob["p"][:1], prob["k_ff"][:1])), gp.device)
        row = {}
        for mode, (mb, di) in (("sync", (False, False)), ("mailbox", (True, False)), ("direct", (True, True))):
            io["mailbox"], io["direct"] = mb, di
            row[mode] = (t(lambda: gp(prob["p"][:1], prob["k_ff"][:1])),
                         t(lambda: gp.linearize_predict(prob["p"][:1], prob["k_ff"][:1], True)))
            if mode == "direct" and not io["direct"]:
                row[mode] = (float("nan"), float("nan"))
        kern = t(lambda: gp.predict_device(x, True))
        row["server"] = (float("nan"), float("nan"))
        if gp.start_server(idle_timeout_s=0.05):
            # (no device-wide synchronisation around these loops: it would wait for the resident kernel's idle time-out)
            row["server"] = (t_host(lambda: gp(prob["p"][:1], prob["k_ff"][:1])),
                             t_host(lambda: gp.linearize_predict(prob["p"][:1], prob["k_ff"][:1], True)))
            gp.stop_server()
        print("n_s=%d N=%5d  __call__: copy+sync %.1f, mailbox %.1f, one command %.1f, resident server %.1f || "
              "linearize_predict(jacobians=True): %.1f, %.1f, %.1f, server %.1f || kernel alone (async) %.1f" % (
                  n_s, N, row["sync"][0], row["mailbox"][0], row["direct"][0], row["server"][0], row["sync"][1],
                  row["mailbox"][1], row["direct"][1], row["server"][1], kern), flush=True)
        del gp
    # the CPU columns last: the BLAS pool keeps its threads spinning after a call and would disturb the GPU timings
    # (one BLAS thread and all of them: a pool of hundreds of threads costs a 100-point model more than it brings)
    from threadpoolctl import threadpool_limits
    for n_s, N in SIZES:
        prob = workload.make_problem(9, N, n_s, 1, 4, sf2=0.01)
        reps = 200 if N <= 1000 else 20
        with threadpool_limits(limits=1):
            one = (cpu_time(prob, False, reps), cpu_time(prob, True, reps))
        allc = (cpu_time(prob, False, reps), cpu_time(prob, True, reps))
        print("n_s=%d N=%5d  CPU NumPy (oracle): __call__ %.1f us with 1 BLAS thread, %.1f with all %d cores; "
              "linearize_predict(jacobians=True) %.1f / %.1f us" % (n_s, N, one[0], allc[0], os.cpu_count() or 1, one[1], allc[1]),
              flush=True)


if __name__ == "__main__":
    if "--kern" in sys.argv:
        main_kern(sys.argv[sys.argv.index("--kern") + 1])
    else:
        main()
