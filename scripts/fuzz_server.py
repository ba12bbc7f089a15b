#!/usr/bin/env python3
"""Randomised sweep of the blocking single-query surface through the RESIDENT server (sr_gp_server_*): random model
sizes across the padded sizes 128 ... 512 (one workgroup per output up to 128 rows, Np / 64 parts beyond), 1 ... 4 outputs,
D = 2 ... 5, the four kernel identifiers (ARD-RBF, mat52, lin_rbf, lin_mat52), with appends, refits and idle time-outs in between -- against the CPU oracle's closed forms and against the
launched routes of the same model.  Run on the GPU box:
    python scripts/fuzz_server.py [cases] [seed]
Exits non-zero on the first case outside the tolerances of the parity tests."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as orc                      # noqa: E402  (checker)
from _helpers import hip_model, oracle_model, mu_atol    # noqa: E402


def check(gp, om, syn, t, worst, n_s):
    p, k = syn["p"][t:t + 1], syn["k_ff"][t:t + 1]
    x = np.hstack((p, k))
    mu, sig, jac = gp(p, k)
    lin = gp.linearize_predict(p, k, True)
    rmu, rvar, rjac = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
    rjv, rhm = orc.gp_linearize_extras(x[0], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    at = max(mu_atol(om), 1e-12)
    sf = float(np.max(om["signal_var"]))
    e = {"mu": np.abs(mu[:, 0] - rmu[0]).max() / (at + 1e-9 * np.abs(rmu).max()),
         "var": np.abs(lin[1][:, 0] - rvar[0]).max() / (1e-9 * sf),
         "jac": np.abs(jac - rjac[0]).max() / (10 * at + 1e-9 * np.abs(rjac).max()),
         "jvar": np.abs(lin[3] - rjv).max() / (1e-7 * np.abs(rjv).max() + 1e-9 * max(1.0, np.abs(rjv).max())),
         "hess": np.abs(lin[4] - rhm).max() / (1e-8 * np.abs(rhm).max() + 1e3 * at)}
    for kk, v in e.items():
        worst[kk] = max(worst[kk], float(v))
    return max(e.values())


class KernCase(object):
    """a model with one of the journal kernels (mat52 / lin_rbf / lin_mat52) and the oracle's closed forms for it"""

    def __init__(self, kt, rng, Z, Y, n_s, n_u):
        from safe_exploration_amd import SimpleGPModel
        self.kt, self.n_s, self.n_u = kt, n_s, n_u
        self.hyp = [orc.make_hyp(kt, rng, n_s + n_u) for _ in range(n_s)]
        self.noise = np.full(n_s, 0.02)
        self.gp = SimpleGPModel(n_s, n_s, n_u, kern_types=[kt] * n_s,
                                hyp=[dict(h, noise_variance=nv) for h, nv in zip(self.hyp, self.noise)])
        self.gp.train(Z, Y, opt_hyp=False)
        self.fit(Z, Y)

    def fit(self, Z, Y):
        self.Z = Z
        self.beta, self.inv_K = orc.gp_fit_k(Z, Y, [self.kt] * self.n_s, self.hyp, self.noise + 1e-5)

    def check(self, syn, t, worst):
        kts = [self.kt] * self.n_s
        p, k = syn["p"][t:t + 1], syn["k_ff"][t:t + 1]
        x = np.hstack((p, k))
        mu, sig, jac = self.gp(p, k)
        lin = self.gp.linearize_predict(p, k, True)
        rmu, rvar = orc.gp_predict_k(x, self.Z, self.beta, self.inv_K, kts, self.hyp)
        rjac = orc.gp_mean_jacobian_k(x, self.Z, self.beta, kts, self.hyp)
        rjv, rhm = orc.gp_linearize_extras_k(x[0], self.Z, self.beta, self.inv_K, kts, self.hyp)
        scale = max(np.abs(self.beta).sum(0).max(), 1.0)
        e = {"mu": np.abs(mu[:, 0] - rmu[0]).max() / (1e-11 * scale + 1e-9 * np.abs(rmu).max()),
             "var": np.abs(lin[1][:, 0] - rvar[0]).max() / (1e-8 * max(1.0, float(rvar.max()))),
             "jac": np.abs(jac - rjac[0]).max() / (1e-10 * scale + 1e-9 * np.abs(rjac).max()),
             "jvar": np.abs(lin[3] - rjv).max() / (1e-6 * np.abs(rjv).max() + 1e-8 * max(1.0, np.abs(rjv).max())),
             "hess": np.abs(lin[4] - rhm).max() / (1e-8 * np.abs(rhm).max() + 1e-10 * scale)}
        for kk, v in e.items():
            worst[kk] = max(worst[kk], float(v))
        return max(e.values())


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {"mu": 0.0, "var": 0.0, "jac": 0.0, "jvar": 0.0, "hess": 0.0}
    for c in range(cases):
        n_s, n_u = [(1, 1), (2, 1), (4, 1), (3, 2), (2, 2), (3, 1)][rng.integers(6)]
        N = int(rng.choice([1, 3, 60, 127, 128, 129, 150, 200, 255, 256, 257, 300, 383, 384, 385, 450, 500, 511]))
        extra = int(rng.integers(0, 4))
        syn = orc.make_synthetic(int(rng.integers(1 << 30)), N + extra, n_s, n_u, 8, sf2=float(rng.choice([1.0, 0.01])))
        Z, Y = syn["Z"], syn["Y"]
        kt = str(rng.choice(["rbf", "rbf", "mat52", "lin_rbf", "lin_mat52", "lin_mat52"]))
        if kt == "rbf":
            gp = hip_model(Z[:N], Y[:N], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
            om = oracle_model(Z[:N], Y[:N], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
            chk = lambda t: check(gp, om, syn, t, worst, n_s)
        else:
            kc = KernCase(kt, rng, Z[:N], Y[:N], n_s, n_u)
            gp = kc.gp
            chk = lambda t: kc.check(syn, t, worst)
        gp.append_limit = 10 ** 9
        armed = gp.start_server(idle_timeout_s=float(rng.choice([0.0005, 0.002, 0.05])))
        err = 0.0
        for t in range(4):
            err = max(err, chk(t))
            if rng.integers(3) == 0:
                time.sleep(0.003)                            # the server may leave on its idle time-out in between
        for i in range(N, N + extra):                        # the model grows under the armed server (may cross a padded size)
            gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
        if extra:
            if kt == "rbf":
                om = oracle_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"])
            else:
                kc.fit(Z, Y)
            for t in range(4, 8):
                err = max(err, chk(t))
        a, r, nl, nc = gp.server_state()
        status = "ok" if err <= 1.0 and armed and (nc > 0 or not a) else "FAIL"
        print("%3d %-9s N=%3d+%d n_s=%d n_u=%d armed=%d launches=%d calls=%d  worst err/tol %.2e  %s" %
              (c, kt, N, extra, n_s, n_u, int(a), nl, nc, err, status), flush=True)
        if status != "ok":
            sys.exit(1)
        del gp
    print("worst error / tolerance:", worst)


if __name__ == "__main__":
    main()
