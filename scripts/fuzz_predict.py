#!/usr/bin/env python3
"""Randomised sweep of the prediction entry points across every dispatch path (one-launch small-model pass,
streaming T <= 16, 64-tiles, split-K, plain MFMA tiles) against the CPU oracle.  Run on the GPU box:
    python scripts/fuzz_predict.py [cases] [seed]
Exits non-zero on the first case outside the SURVEY 8(d) tolerances."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as orc                      # noqa: E402  (checker)
from _helpers import hip_model, oracle_model, mu_atol    # noqa: E402
from safe_exploration_amd import gp_reachability as reach  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {"mu": 0.0, "var": 0.0, "jac": 0.0, "q": 0.0}
    for c in range(cases):
        n_s, n_u = [(2, 1), (4, 1), (3, 2), (5, 4), (8, 4)][rng.integers(5)]
        N = int(rng.choice([1, 2, 17, 100, 127, 128, 129, 200, 255, 256, 257, 300, 511, 640, 1100, 1300]))
        T = int(rng.choice([1, 2, 5, 15, 16, 17, 63, 64, 65, 127, 128, 129, 300, 1024, 1025, 1500]))
        syn = orc.make_synthetic(int(rng.integers(1 << 30)), N, n_s, n_u, T, sf2=float(rng.choice([1.0, 0.01])))
        gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
        om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        x = np.hstack((syn["p"], syn["k_ff"]))
        mu, var, jac = gp.predict(x, None, True)
        rmu, rvar, rjac = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
        at = max(mu_atol(om), 1e-12)
        e_mu = np.abs(mu - rmu).max() / (at + 1e-9 * np.abs(rmu).max())
        e_jac = np.abs(jac - rjac).max() / (10 * at + 1e-9 * np.abs(rjac).max())
        e_var = np.abs(var - rvar).max() / (1e-9 * float(np.max(syn["signal_var"])))
        l = np.full(n_s, 0.05 if n_s <= 4 else 0.01)
        p1, q1 = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0)
        rp, rq, _ = orc.onestep_reachability_vectorised(om, syn["p"], syn["Q"], syn["k_ff"], syn["k_fb"], l, l, 2.0,
                                                        np.eye(n_s), np.zeros((n_s, n_u)))
        e_q = np.abs(q1 - rq).max() / (1e-8 * np.abs(rq).max() + 1e-14)
        for k, v in (("mu", e_mu), ("var", e_var), ("jac", e_jac), ("q", e_q)):
            worst[k] = max(worst[k], float(v))
        status = "ok" if max(e_mu, e_var, e_jac, e_q) <= 1.0 else "FAIL"
        print("%3d N=%4d T=%4d n_s=%d n_u=%d  err/tol mu %.2e var %.2e jac %.2e Q %.2e  %s" %
              (c, N, T, n_s, n_u, e_mu, e_var, e_jac, e_q, status), flush=True)
        if status != "ok":
            sys.exit(1)
    print("worst error / tolerance:", worst)


if __name__ == "__main__":
    main()
