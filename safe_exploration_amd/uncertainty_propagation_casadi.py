"""Gaussian moment propagation through the GP dynamics (the CautiousMPC baseline), numeric and batched.

Numeric counterparts of the symbolic graph builders in
/root/reference/safe_exploration/uncertainty_propagation_casadi.py (one_step_taylor :11-87,
multi_step_taylor_symbolic :88-146, mean_equivalent_multistep :149-207, one_step_mean_equivalent
:210-283) with the same argument orders.  The covariance algebra of both schemes collapses to
``Sigma_new = H Sigma H^T + diag(var)`` with ``H = a + J_x + (b + J_u) K`` (Taylor) or ``H = a + b K``
(mean-equivalent); it runs in the same per-query kernel as the robust ellipsoid step.
The input transform ``a_gp_inp_x`` of the reference (the GP sees ``a_gp_inp_x @ state``, :40-47,60) is supported:
the state Jacobian is chain-ruled through the constant matrix.
"""
import numpy as np

from . import _buffers as B
from ._lib import lib, check
from .ssm_hip.gaussian_process import SimpleGPModel

TAYLOR, MEAN_EQUIVALENT = 1, 2


def _lin(a, b, n_s, n_u):
    if a is None:
        a, b = np.eye(n_s), np.zeros((n_s, n_u))
    return np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)


def multistep_moments_batch(mu_0, ssm, k_ff, k_fb, a=None, b=None, mode=TAYLOR, a_gp_inp_x=None):
    """T trajectories x H steps.  mu_0 (T,n_s); k_ff (T,H,n_u); k_fb (T,H-1,n_u,n_s); a_gp_inp_x (n_x_in,n_s) or None.
    Returns mu_all (T,H,n_s), sigma_all (T,H,n_s,n_s), gp_var_all (T,H,n_s)."""
    from .gp_reachability import _input_transform, _reach_dims
    if not isinstance(ssm, SimpleGPModel):
        raise TypeError("multistep_moments_batch needs the HIP SimpleGPModel")
    ssm._need_trained()
    as_t = B.is_tensor(mu_0)
    hd = ssm._handle
    dev = hd.device
    n_s, n_u = _reach_dims(hd, a_gp_inp_x)
    m0 = B.as_dev(mu_0, dev)
    T = m0.shape[0]
    kff = B.as_dev(k_ff, dev)
    if kff.dim() != 3 or kff.shape[0] != T or kff.shape[2] != n_u:
        raise ValueError("k_ff must be (T, H, {})".format(n_u))
    H = kff.shape[1]
    kfb = B.as_dev(k_fb, dev, (T, H - 1, n_u, n_s)) if H > 1 else None
    a, b = _lin(a, b, n_s, n_u)
    ta, tb = B.const_dev(a, dev, (n_s, n_s)), B.const_dev(b, dev, (n_s, n_u))
    mu_all, sigma_all = B.empty((T, H, n_s), dev), B.empty((T, H, n_s, n_s), dev)
    var_all = B.empty((T, H, n_s), dev)
    with _input_transform(ssm, a_gp_inp_x):
        check(lib.sr_multistep_moments(hd.h, T, H, int(mode), B.ptr(m0), B.ptr(kff), B.ptr(kfb), B.ptr(ta), B.ptr(tb),
                                       B.ptr(mu_all), B.ptr(sigma_all), B.ptr(var_all), B.stream_ptr(dev)))
    outs = (mu_all, sigma_all, var_all)
    if as_t:
        return outs
    from .gp_reachability import _raise_if_chain_failed
    _raise_if_chain_failed(hd, dev, False)
    return tuple(B.to_numpy(o) for o in outs)


def moment_step_batch(mu_x, k_ff, mu_g, var_g, jac_g, sigma_x=None, k_fb=None, a=None, b=None, mode=TAYLOR,
                      device=None):
    """One propagation step for T inputs with caller-supplied GP outputs (any StateSpaceModel)."""
    as_t = B.is_tensor(mu_x)
    dev = B.resolve_device(mu_x.device if as_t else device)
    mx = B.as_dev(mu_x, dev)
    T, n_s = mx.shape
    kff = B.as_dev(k_ff, dev).reshape(T, -1)
    n_u = kff.shape[1]
    sx = B.as_dev(sigma_x, dev, (T, n_s, n_s)) if sigma_x is not None else None
    if sx is not None and k_fb is None:
        raise ValueError("k_fb is required when sigma_x is given")
    kfb = B.as_dev(k_fb, dev, (T, n_u, n_s)) if sx is not None else None
    tmu, tvar = B.as_dev(mu_g, dev, (T, n_s)), B.as_dev(var_g, dev, (T, n_s))
    tjac = B.as_dev(jac_g, dev, (T, n_s, n_s + n_u)) if (jac_g is not None and sx is not None) else None
    a, b = _lin(a, b, n_s, n_u)
    ta, tb = B.const_dev(a, dev, (n_s, n_s)), B.const_dev(b, dev, (n_s, n_u))
    mo, so = B.empty((T, n_s), dev), B.empty((T, n_s, n_s), dev)
    check(lib.sr_moment_step(dev.index, T, n_s, n_u, int(mode), B.ptr(mx), B.ptr(sx), B.ptr(kff), B.ptr(kfb),
                             B.ptr(tmu), B.ptr(tvar), B.ptr(tjac), B.ptr(ta), B.ptr(tb), B.ptr(mo), B.ptr(so),
                             B.stream_ptr(dev)))
    return (mo, so) if as_t else (B.to_numpy(mo), B.to_numpy(so))


def _one_step(mu_x, ssm, k_ff, sigma_x, k_fb, a, b, a_gp_inp_x, mode):
    mu_x = np.asarray(mu_x, dtype=np.float64)
    k_ff = np.asarray(k_ff, dtype=np.float64)
    n_s = mu_x.shape[0]
    t = None if a_gp_inp_x is None else np.asarray(a_gp_inp_x, dtype=np.float64)
    x_bar = mu_x if t is None else t.dot(mu_x)
    out = ssm(x_bar.T, k_ff.T)                         # (mu n x 1, var n x 1, jac n x D)
    mu_g, var_g = np.asarray(out[0], dtype=np.float64), np.asarray(out[1], dtype=np.float64)
    jac = np.asarray(out[2], dtype=np.float64) if sigma_x is not None else None
    if jac is not None and t is not None:              # chain rule through the constant input map (:60)
        jac = np.hstack((jac[:, :t.shape[0]].dot(t), jac[:, t.shape[0]:]))
    jac = jac[None] if jac is not None else None
    sx = None if sigma_x is None else np.asarray(sigma_x, dtype=np.float64)[None]
    kfb = None if sigma_x is None else np.asarray(k_fb, dtype=np.float64)[None]
    mu_new, sigma_new = moment_step_batch(mu_x.T, k_ff.T, mu_g.reshape(1, n_s), var_g.reshape(1, n_s), jac, sx,
                                          kfb, a, b, mode)
    return mu_new.reshape(n_s, 1), sigma_new[0], var_g.reshape(1, n_s)


def one_step_taylor(mu_x, ssm, k_ff, sigma_x=None, k_fb=None, a=None, b=None, a_gp_inp_x=None):
    """First-order Taylor propagation of N(mu_x, sigma_x) (uncertainty_propagation_casadi.py:11-87).
    Returns mu_new (n_s,1), sigma_new (n_s,n_s), gp variances (1,n_s)."""
    return _one_step(mu_x, ssm, k_ff, sigma_x, k_fb, a, b, a_gp_inp_x, TAYLOR)


def one_step_mean_equivalent(mu_x, ssm, k_ff, sigma_x=None, k_fb=None, a=None, b=None, a_gp_inp_x=None):
    """'Mean-equivalent' propagation: no Jacobian cross terms (uncertainty_propagation_casadi.py:210-283)."""
    return _one_step(mu_x, ssm, k_ff, sigma_x, k_fb, a, b, a_gp_inp_x, MEAN_EQUIVALENT)


def _multi(mu_0, ssm, k_ff, k_fb, sigma_0, a, b, a_gp_inp_x, mode):
    if sigma_0 is not None:
        raise NotImplementedError("Still need  to do this")        # like the reference (:124, :170)
    k_ff = np.asarray(k_ff, dtype=np.float64)
    T, n_u = k_ff.shape
    n_s = np.shape(mu_0)[0]
    kfb = np.asarray(k_fb, dtype=np.float64).reshape(T - 1, n_u, n_s)[None] if T > 1 else None
    if isinstance(ssm, SimpleGPModel):
        mu_all, sigma_all, var_all = multistep_moments_batch(np.asarray(mu_0, dtype=np.float64).reshape(1, n_s),
                                                             ssm, k_ff[None], kfb, a, b, mode, a_gp_inp_x)
        return mu_all[0], sigma_all[0].reshape(T, n_s * n_s), var_all[0]
    one = one_step_taylor if mode == TAYLOR else one_step_mean_equivalent
    mu_new, sigma_new, gv = one(mu_0, ssm, k_ff[0].reshape(n_u, 1), None, None, a, b, a_gp_inp_x)
    mus, sigmas, gvs = [mu_new.T], [sigma_new.reshape(1, -1)], [gv]
    for i in range(T - 1):
        mu_new, sigma_new, gv = one(mu_new, ssm, k_ff[i + 1].reshape(n_u, 1), sigma_new, kfb[0, i], a, b, a_gp_inp_x)
        mus.append(mu_new.T), sigmas.append(sigma_new.reshape(1, -1)), gvs.append(gv)
    return np.vstack(mus), np.vstack(sigmas), np.vstack(gvs)


def multi_step_taylor(mu_0, ssm, k_ff, k_fb, sigma_0=None, a=None, b=None, a_gp_inp_x=None):
    """Numeric ``multi_step_taylor_symbolic`` (uncertainty_propagation_casadi.py:88-146): k_ff (T,n_u),
    k_fb list/array of T-1 (n_u,n_s) gains.  Returns mu_all (T,n_s), sigma_all (T,n_s*n_s), gp variances (T,n_s)."""
    return _multi(mu_0, ssm, k_ff, k_fb, sigma_0, a, b, a_gp_inp_x, TAYLOR)


def mean_equivalent_multistep(mu_0, ssm, k_ff, k_fb, sigma_0=None, a=None, b=None, a_gp_inp_x=None):
    """uncertainty_propagation_casadi.py:149-207."""
    return _multi(mu_0, ssm, k_ff, k_fb, sigma_0, a, b, a_gp_inp_x, MEAN_EQUIVALENT)


multi_step_taylor_symbolic = multi_step_taylor      # the reference's name (uncertainty_propagation_casadi.py:11)
