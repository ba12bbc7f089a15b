"""Shared helpers for the parity tests (oracle side is TEST infrastructure)."""
import numpy as np

from oracle import oracle_np as orc


def hyp_from(ls, sf2, noise_var, noise_diag=1e-5):
    """hyp list for SimpleGPModel such that its diagonal term equals oracle's noise_var + 1e-8:
    oracle.gp_fit receives noise_var (already incl. noise_diag) and adds the GPy jitter itself."""
    return [{"lengthscale": ls[d], "variance": sf2[d], "noise_variance": noise_var[d] - noise_diag}
            for d in range(len(sf2))]


def oracle_model(Z, Y, ls, sf2, noise_var):
    beta, inv_K, chol = orc.gp_fit(Z, Y, ls, sf2, noise_var)
    return dict(Z=Z, beta=beta, inv_K=inv_K, chol=chol, lengthscale=ls, signal_var=sf2)


def hip_model(Z, Y, ls, sf2, noise_var, n_s, n_u):
    from safe_exploration_amd import SimpleGPModel
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=hyp_from(ls, sf2, noise_var))
    gp.train(Z, Y, opt_hyp=False)
    return gp


def mu_atol(model):
    """atol for mu / jac: 1e-12 * sigma_f * |alpha|_1  (SURVEY 8d)."""
    return 1e-12 * float(np.sqrt(np.max(model["signal_var"])) * np.abs(model["beta"]).sum(0).max())


_ORACLE_CACHE = {}


def cached_oracle_model(seed, N, n_s, n_u, sf2=1.0):
    """oracle fit of orc.make_synthetic(seed, N, ...) -- O(N^3) on the CPU, shared between the full-size tests."""
    key = (seed, N, n_s, n_u, sf2)
    if key not in _ORACLE_CACHE:
        syn = orc.make_synthetic(seed, N, n_s, n_u, 4, sf2=sf2)
        _ORACLE_CACHE[key] = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    return _ORACLE_CACHE[key]
