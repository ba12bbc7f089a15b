"""Synthetic workload generators for the benchmark / scaling runs.

``random_rollout_controls`` reproduces the batch generator of the reference's
uncertainty_propagation_runner (/root/reference/safe_exploration/uncertainty_propagation_runner.py:32-34:
k_fb = .1 randn(n_safe-1, n_u, n_s), k_ff = .1 randn(n_safe, n_u)) with a leading batch axis.
"""
import numpy as np


def make_problem(seed, N, n_s, n_u, T, noise=1e-2, sf2=1.0):
    """Seeded synthetic GP-dynamics problem + T one-step query states.

    Z ~ U[-1,1]^(N x D); y_d = sf (sin(2 z.w_d) + 0.05 randn); lengthscale ~ U[0.5,1.5];
    signal variance sf2; noise variance noise*sf2 (+1e-5 noise_diag added by SimpleGPModel.train).
    Queries: p ~ 0.3 randn, k_ff ~ 0.1 randn, k_fb ~ 0.1 randn, Q = 0.01 A A^T + 0.01 I.
    """
    rng = np.random.default_rng(seed)
    D = n_s + n_u
    Z = rng.uniform(-1, 1, (N, D))
    Wd = rng.standard_normal((n_s, D))
    Y = np.sqrt(sf2) * (np.sin(2.0 * Z.dot(Wd.T)) + 0.05 * rng.standard_normal((N, n_s)))
    ls = rng.uniform(0.5, 1.5, (n_s, D))
    prob = dict(Z=Z, Y=Y, lengthscale=ls, signal_var=np.full(n_s, float(sf2)),
                noise_var=np.full(n_s, noise * sf2))
    prob.update(make_queries(seed + 7919, n_s, n_u, T))
    return prob


def make_queries(seed, n_s, n_u, T):
    rng = np.random.default_rng(seed)
    p = 0.3 * rng.standard_normal((T, n_s))
    k_ff = 0.1 * rng.standard_normal((T, n_u))
    k_fb = 0.1 * rng.standard_normal((T, n_u, n_s))
    A = rng.standard_normal((T, n_s, n_s))
    Q = 0.01 * np.einsum('tij,tkj->tik', A, A) + 0.01 * np.eye(n_s)[None]
    return dict(p=p, k_ff=k_ff, k_fb=k_fb, Q=Q)


def random_rollout_controls(seed, T, H, n_s, n_u, p0_std=0.1):
    """T random rollouts of horizon H: p0 ~ N(0, p0_std^2), k_ff (T,H,n_u), k_fb (T,H-1,n_u,n_s)."""
    rng = np.random.default_rng(seed)
    return dict(p0=p0_std * rng.standard_normal((T, n_s)),
                k_ff=0.1 * rng.standard_normal((T, H, n_u)),
                k_fb=0.1 * rng.standard_normal((T, max(H - 1, 0), n_u, n_s)))


def hyp_list(prob):
    """hyp argument of SimpleGPModel for a problem dict."""
    return [{"lengthscale": prob["lengthscale"][d], "variance": prob["signal_var"][d],
             "noise_variance": prob["noise_var"][d]} for d in range(len(prob["signal_var"]))]
