#!/usr/bin/env python3
"""Randomised sweep of the multi-step chains of small models (persistent kernel and per-step launches) and of row-append
sequences against the CPU oracle / a refit.  GPU box:  python scripts/fuzz_chain.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as orc                      # noqa: E402  (checker)
from _helpers import hip_model, oracle_model             # noqa: E402
from safe_exploration_amd import gp_reachability as reach  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {"p": 0.0, "q": 0.0, "append_var": 0.0}
    taken = 0
    for c in range(cases):
        n_s, n_u = [(1, 1), (2, 1), (3, 1), (4, 1), (2, 2), (3, 2)][rng.integers(6)]
        N = int(rng.choice([20, 100, 128, 129, 200, 256, 300, 384, 400, 512]))
        T = int(rng.choice([1, 3, 16, 17, 100, 240, 480, 481, 1000]))
        H = int(rng.choice([1, 2, 3, 5, 9]))
        syn = orc.make_synthetic(int(rng.integers(1 << 30)), N, n_s, n_u, T, sf2=float(rng.choice([1.0, 0.01])))
        gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
        om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        k_ff = 0.3 * rng.standard_normal((T, H, n_u))
        k_fb = 0.1 * rng.standard_normal((T, max(H - 1, 0), n_u, n_s))
        l_mu, l_sg = np.linspace(0.005, 0.008, n_s), np.linspace(0.002, 0.003, n_s)
        a = 0.6 * np.eye(n_s) + 0.03 * rng.standard_normal((n_s, n_s))
        b = 0.1 * rng.standard_normal((n_s, n_u))
        with_q0 = bool(rng.integers(2))
        q0 = syn["Q"] if with_q0 else None
        kfb0 = syn["k_fb"] if with_q0 else None
        if H == 1:
            p1, q1 = reach.onestep_reachability_batch(syn["p"], gp, k_ff[:, 0], l_mu, l_sg, q0, kfb0, 2.0, a, b)
            rp, rq, _ = orc.onestep_reachability_batch(om, syn["p"], q0, k_ff[:, 0], kfb0, l_mu, l_sg, 2.0, a, b)
            p_all, q_all, rp, rq = p1[:, None], q1[:, None], rp[:, None], rq[:, None]
        else:
            p_all, q_all = reach.multistep_reachability_batch(syn["p"], gp, k_fb, k_ff, l_mu, l_sg, q0, 2.0, a, b, kfb0)
            n = min(T, 64)
            rp, rq = orc.multistep_reachability_batch(om, syn["p"][:n], k_fb[:n], k_ff[:n], l_mu, l_sg,
                                                      None if q0 is None else q0[:n], 2.0, a, b,
                                                      None if kfb0 is None else kfb0[:n])
            p_all, q_all = p_all[:n], q_all[:n]
        taken += int(gp.last_chain)
        e_p = np.abs(p_all - rp).max() / (1e-10 + 1e-7 * np.abs(rp).max())
        e_q = np.abs(q_all - rq).max() / (1e-12 + 1e-6 * np.abs(rq).max())
        # a random append sequence on the same model against a refit
        adds = [int(v) for v in rng.choice([1, 2, 5, 16, 17, 40], size=int(rng.integers(1, 4)))]
        extra = orc.make_synthetic(int(rng.integers(1 << 30)), sum(adds), n_s, n_u, 4, sf2=1.0)
        Zx = syn["Z"][rng.integers(0, N, sum(adds))] + 0.05 * extra["Z"][:, :syn["Z"].shape[1]]
        Yx = extra["Y"]
        lo = 0
        gp.append_limit = 10 ** 9           # exercise the append kernels whatever the size policy would choose
        for m in adds:
            gp.update_model(Zx[lo:lo + m], Yx[lo:lo + m], opt_hyp=False, replace_old=False)
            lo += m
        full = hip_model(np.vstack((syn["Z"], Zx)), np.vstack((syn["Y"], Yx)), syn["lengthscale"], syn["signal_var"],
                         syn["noise_var"], n_s, n_u)
        x = np.hstack((syn["p"][:16], syn["k_ff"][:16]))
        va, vf = gp.predict(x)[1], full.predict(x)[1]
        e_v = np.abs(va - vf).max() / (1e-9 * float(np.max(syn["signal_var"])))
        ok = e_p <= 1 and e_q <= 1 and e_v <= 1
        print("%3d n_s=%d n_u=%d N=%4d T=%4d H=%d q0=%d chain=%d adds=%-12s err/tol p %.2e q %.2e append var %.2e  %s" % (
            c, n_s, n_u, N, T, H, with_q0, gp.last_chain, adds, e_p, e_q, e_v, "ok" if ok else "FAIL"), flush=True)
        for k, v in (("p", e_p), ("q", e_q), ("append_var", e_v)):
            worst[k] = max(worst[k], float(v))
        if not ok:
            sys.exit(1)
        del gp, full
    print("persistent kernel taken in %d of %d cases; worst error / tolerance: %s" % (taken, cases, worst))


if __name__ == "__main__":
    main()
