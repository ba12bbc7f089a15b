"""Ellipsoidal one-step / multi-step reachability on MI355X.

Same function names, positional orders and return conventions as
/root/reference/safe_exploration/gp_reachability.py:19-250; the ``*_batch`` variants carry a
leading T axis (T query states / T candidate trajectories) and are what the benchmark measures.
All arithmetic happens in the HIP kernels behind include/safereach.h:

  * ssm is a HIP ``SimpleGPModel``  -> sr_onestep_reach / sr_multistep_reach (GP + ellipsoid fused
    on the device, nothing returns to the host between the kernels),
  * any other StateSpaceModel     -> its ``ssm(x, u)`` is called on the host exactly like the
    reference does (gp_reachability.py:74,101) and the ellipsoid algebra runs in sr_ellipsoid_step.
"""
import ctypes

import numpy as np
import torch

from . import _buffers as B
from ._lib import lib, check
from .ssm_hip.gaussian_process import SimpleGPModel
from .utils import print_ellipsoid


def _lin_model(a, b, n_s, n_u):
    if a is None:                      # gp_reachability.py:61-63
        a = np.eye(n_s)
        b = np.zeros((n_s, n_u))
    return np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)


class _input_transform(object):
    """``with _input_transform(ssm, t_z_gp):`` -- the GP is evaluated at t_z_gp @ state for the reachability calls
    inside the block (gp_reachability_casadi.py:60-61,85,94-97); restored to the identity afterwards."""

    def __init__(self, ssm, t_z_gp):
        self.hd, self.t = ssm._handle, t_z_gp

    def __enter__(self):
        if self.t is not None:
            hd = self.hd
            t = np.asarray(self.t, dtype=np.float64)
            if t.ndim != 2 or t.shape[1] != hd.n_out or not (1 <= t.shape[0] < hd.D):
                raise ValueError("t_z_gp must be (n_x_in, {}) with n_x_in < D = {}".format(hd.n_out, hd.D))
            self.dev_t = B.as_dev(t, hd.device)
            check(lib.sr_gp_set_input_transform(hd.h, B.ptr(self.dev_t), int(t.shape[0]), B.stream_ptr(hd.device)))
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            check(lib.sr_gp_set_input_transform(self.hd.h, None, 0, B.stream_ptr(self.hd.device)))
        return False


def _reach_dims(hd, t_z_gp):
    """(n_s, n_u) of the reachability problem on this model: D = n_x_in + n_u with n_x_in = rows of t_z_gp"""
    n_xin = hd.n_out if t_z_gp is None else int(np.shape(t_z_gp)[0])
    if hd.D - n_xin < 1:
        raise ValueError("the model has {} inputs for {} states: a GP input transform (t_z_gp / a_gp_inp_x, "
                         "(n_x_in, n_s) with n_x_in < {}) is needed".format(hd.D, hd.n_out, hd.D))
    return hd.n_out, hd.D - n_xin


def _raise_if_bad(n_bad):
    if n_bad is not None and int(n_bad.item()) > 0:
        # the reference asserts inside ellipsoid_from_rectangle (utils_ellipsoid.py:226-228)
        raise AssertionError("all elements of u_b need to be greater than zero! "
                             "({} query state(s) affected)".format(int(n_bad.item())))


def chain_timed_out(hd, sync_device=None):
    """True if a persistent multi-step launch of this handle failed since the last query: its workgroups did not become
    co-resident within 100 ms and it filled its outputs with NaN (``sr_gp_chain_status``).  Synchronises the current
    stream first when a device is given -- the status word is only final once the launch has ended."""
    if sync_device is not None:
        torch.cuda.current_stream(sync_device).synchronize()
    flag = ctypes.c_int(0)
    check(lib.sr_gp_chain_status(hd.h, ctypes.byref(flag)))
    return bool(flag.value)


def _raise_if_chain_failed(hd, dev, synced):
    """After a call whose results the caller is about to read on the host: never hand NaN ellipsoids back silently."""
    if lib.sr_gp_last_chain(hd.h) and chain_timed_out(hd, None if synced else dev):
        raise RuntimeError("libsafereach: the persistent multi-step kernel timed out (its workgroups were not co-resident "
                           "within 100 ms: the device was busy with other work); the model now uses per-step launches -- "
                           "repeat the call")


def onestep_reachability_batch(p_center, ssm, k_ff, l_mu, l_sigma, q_shape=None, k_fb=None,
                               c_safety=1., a=None, b=None, check_bounds=False, return_var=False, t_z_gp=None):
    """Batched ``onestep_reachability``.

    p_center (T,n_s); k_ff (T,n_u); q_shape (T,n_s,n_s) or None; k_fb (T,n_u,n_s) or None.
    numpy in -> numpy out, torch (device) in -> torch out.  Requires a HIP ``SimpleGPModel``.
    t_z_gp (n_x_in, n_s): the GP's state inputs are t_z_gp @ state (the argument of the same name of
    gp_reachability_casadi.onestep_reachability, :17-19,60-61); the model then has D = n_x_in + n_u inputs.
    Returns p_new (T,n_s), q_new (T,n_s,n_s) [, var (T,n_s)].
    """
    if not isinstance(ssm, SimpleGPModel):
        raise TypeError("onestep_reachability_batch needs the HIP SimpleGPModel")
    as_t = B.is_tensor(p_center)
    hd = ssm._handle
    ssm._need_trained()
    dev = hd.device
    n_s, n_u = _reach_dims(hd, t_z_gp)
    if q_shape is not None and k_fb is None:
        raise ValueError("k_fb is required when q_shape is given")
    staged = (not as_t) and not any(B.is_tensor(x) for x in (k_ff, q_shape, k_fb))
    if staged:
        np_p = np.ascontiguousarray(np.asarray(p_center, dtype=np.float64))
        staged = np_p.ndim == 2 and np_p.shape[0] * (n_s * n_s + n_s + n_u * n_s + n_u) <= B.STAGING_MAX_DOUBLES
    if staged:
        # NumPy in / out: every argument through one pinned block, every result back through one (B.Staging)
        f64 = lambda x: None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        T = np_p.shape[0]
        if np_p.shape[1] != n_s:
            raise ValueError("p_center must be (T, {})".format(n_s))
        np_q = f64(q_shape).reshape(T, n_s, n_s) if q_shape is not None else None
        np_kfb = f64(k_fb).reshape(T, n_u, n_s) if (k_fb is not None and q_shape is not None) else None
        st = getattr(hd, "_staging", None)
        if st is None:
            st = hd._staging = B.Staging(dev)
        shapes = [(T, n_s), (T, n_s, n_s), (1,)] + ([(T, n_s)] if return_var else [])
        (p, kff, q, kfb), outs_d = st.stage([np_p, f64(k_ff).reshape(T, n_u), np_q, np_kfb], shapes, zero_copy=True)
        p_out, q_out, bad_slot = outs_d[:3]
        var = outs_d[3] if return_var else None
        n_bad = None
        if check_bounds:
            bad_slot.zero_()
            n_bad = bad_slot.view(torch.int32)
    else:
        p = B.as_dev(p_center, dev)
        T = p.shape[0]
        if p.dim() != 2 or p.shape[1] != n_s:
            raise ValueError("p_center must be (T, {})".format(n_s))
        kff = B.as_dev(k_ff, dev, (T, n_u))
        q = B.as_dev(q_shape, dev, (T, n_s, n_s)) if q_shape is not None else None
        kfb = B.as_dev(k_fb, dev, (T, n_u, n_s)) if (k_fb is not None and q is not None) else None
        p_out = B.empty((T, n_s), dev)
        q_out = B.empty((T, n_s, n_s), dev)
        var = B.empty((T, n_s), dev) if return_var else None
        n_bad = B.zeros_i32(1, dev) if check_bounds else None
    a, b = _lin_model(a, b, n_s, n_u)
    ta, tb = B.const_dev(a, dev, (n_s, n_s)), B.const_dev(b, dev, (n_s, n_u))
    tlm, tls = B.const_dev(l_mu, dev, (n_s,)), B.const_dev(l_sigma, dev, (n_s,))
    with _input_transform(ssm, t_z_gp):
        check(lib.sr_onestep_reach(hd.h, T, B.ptr(p), B.ptr(q), B.ptr(kff), B.ptr(kfb), B.ptr(ta),
                                   B.ptr(tb), B.ptr(tlm), B.ptr(tls), float(c_safety), B.ptr(p_out),
                                   B.ptr(q_out), B.ptr(var), B.ptr(n_bad), B.stream_ptr(dev)))
    if staged:
        res = st.fetch()                              # (synchronises the stream)
        if check_bounds:
            n_viol = int(res[2].view(np.int32)[0])
            if n_viol > 0:
                raise AssertionError("all elements of u_b need to be greater than zero! "
                                     "({} query state(s) affected)".format(n_viol))
        _raise_if_chain_failed(hd, dev, True)
        return (res[0], res[1], res[3]) if return_var else (res[0], res[1])
    _raise_if_bad(n_bad)
    outs = (p_out, q_out, var) if return_var else (p_out, q_out)
    if as_t:
        return outs          # device tensors, nothing synchronised: a failed chain is reported by the next entry point
    _raise_if_chain_failed(hd, dev, n_bad is not None)
    return tuple(B.to_numpy(o) for o in outs)


def ellipsoid_step_batch(p_center, k_ff, mu, var, jac, l_mu, l_sigma, q_shape=None, k_fb=None,
                         c_safety=1., a=None, b=None, check_bounds=False, device=None):
    """The ellipsoid algebra of gp_reachability.py:65-156 for T queries whose GP outputs
    (mu (T,n_s), var (T,n_s), jac (T,n_s,n_s+n_u)) are supplied by the caller."""
    as_t = B.is_tensor(p_center)
    dev = B.resolve_device(p_center.device if as_t else device)
    p = B.as_dev(p_center, dev)
    T, n_s = p.shape
    kff = B.as_dev(k_ff, dev).reshape(T, -1)
    n_u = kff.shape[1]
    tmu, tvar = B.as_dev(mu, dev, (T, n_s)), B.as_dev(var, dev, (T, n_s))
    q = B.as_dev(q_shape, dev, (T, n_s, n_s)) if q_shape is not None else None
    if q is not None and (k_fb is None or jac is None):
        raise ValueError("k_fb and jac are required when q_shape is given")
    kfb = B.as_dev(k_fb, dev, (T, n_u, n_s)) if q is not None else None
    tjac = B.as_dev(jac, dev, (T, n_s, n_s + n_u)) if (jac is not None and q is not None) else None
    a, b = _lin_model(a, b, n_s, n_u)
    ta, tb = B.const_dev(a, dev, (n_s, n_s)), B.const_dev(b, dev, (n_s, n_u))
    tlm, tls = B.const_dev(l_mu, dev, (n_s,)), B.const_dev(l_sigma, dev, (n_s,))
    p_out = B.empty((T, n_s), dev)
    q_out = B.empty((T, n_s, n_s), dev)
    n_bad = B.zeros_i32(1, dev) if check_bounds else None
    check(lib.sr_ellipsoid_step(dev.index, T, n_s, n_u, B.ptr(p), B.ptr(q), B.ptr(kff), B.ptr(kfb),
                                B.ptr(tmu), B.ptr(tvar), B.ptr(tjac), B.ptr(ta), B.ptr(tb),
                                B.ptr(tlm), B.ptr(tls), float(c_safety), B.ptr(p_out), B.ptr(q_out),
                                B.ptr(n_bad), B.stream_ptr(dev)))
    _raise_if_bad(n_bad)
    return (p_out, q_out) if as_t else (B.to_numpy(p_out), B.to_numpy(q_out))


def onestep_reachability(p_center, ssm, k_ff, l_mu, l_sigma, q_shape=None, k_fb=None,
                         c_safety=1., verbose=1, a=None, b=None, t_z_gp=None):
    """Overapproximate the reachable set of states under the affine control law u = K(x-p) + k.

    Single-query semantics and argument order of gp_reachability.py:19-156.
    p_center (n_s,1); k_ff (n_u,1); q_shape (n_s,n_s) or None; k_fb (n_u,n_s) or None.
    Returns p_new (n_s,1), q_new (n_s,n_s).
    """
    p_center = np.asarray(p_center, dtype=np.float64)
    k_ff = np.asarray(k_ff, dtype=np.float64)
    n_s = np.shape(p_center)[0]
    n_u = np.shape(k_ff)[0]
    if verbose > 0:
        if q_shape is not None:
            print_ellipsoid(p_center, q_shape, text="initial uncertainty ellipsoid")
        print("\nApplying action:")
        print(k_ff)
    q_b = None if q_shape is None else np.asarray(q_shape, dtype=np.float64)[None]
    kfb_b = None if (k_fb is None or q_shape is None) else np.asarray(k_fb, dtype=np.float64)[None]
    if isinstance(ssm, SimpleGPModel):
        p1, q1 = onestep_reachability_batch(p_center.T, ssm, k_ff.T, l_mu, l_sigma, q_b, kfb_b,
                                            c_safety, a, b, check_bounds=True, t_z_gp=t_z_gp)
    else:
        x_bar = p_center if t_z_gp is None else np.asarray(t_z_gp, dtype=np.float64).dot(p_center)
        out = ssm(x_bar.T, k_ff.T)                     # (mu n x 1, sigma n x 1, jac n x D)
        mu_0, sigm_0 = np.array(out[0], dtype=np.float64), np.array(out[1], dtype=np.float64)
        jac_mu = np.array(out[2], dtype=np.float64) if q_shape is not None else None
        if jac_mu is not None and t_z_gp is not None:   # chain rule through the constant input map
            t = np.asarray(t_z_gp, dtype=np.float64)
            jac_mu = np.hstack((jac_mu[:, :t.shape[0]].dot(t), jac_mu[:, t.shape[0]:]))
        jac_mu = jac_mu[None] if jac_mu is not None else None
        p1, q1 = ellipsoid_step_batch(p_center.T, k_ff.T, mu_0.reshape(1, n_s), sigm_0.reshape(1, n_s),
                                      jac_mu, l_mu, l_sigma, q_b, kfb_b, c_safety, a, b,
                                      check_bounds=True)
    p_1, q_1 = p1.reshape(n_s, 1), q1[0]
    if verbose > 0:
        print_ellipsoid(p_1, q_1, text="accumulated uncertainty current step")
    return p_1, q_1


def multistep_reachability_batch(p_0, gp, k_fb, k_ff, L_mu, L_sigm, q_0=None, c_safety=1., a=None,
                                 b=None, k_fb_init=None, check_bounds=False, t_z_gp=None):
    """Batched ``multistep_reachability``: T independent trajectories, H sequential steps each.

    p_0 (T,n_s); k_fb (T,H-1,n_u,n_s); k_ff (T,H,n_u); q_0 (T,n_s,n_s) or None;
    k_fb_init (T,n_u,n_s) (needed iff q_0 is given).  Returns p_all (T,H,n_s), q_all (T,H,n_s,n_s).
    """
    if not isinstance(gp, SimpleGPModel):
        raise TypeError("multistep_reachability_batch needs the HIP SimpleGPModel")
    gp._need_trained()
    as_t = B.is_tensor(p_0)
    hd = gp._handle
    dev = hd.device
    n_s, n_u = _reach_dims(hd, t_z_gp)
    if q_0 is not None and k_fb_init is None:
        raise ValueError("k_fb_init is required when q_0 is given")
    staged = (not as_t) and not any(B.is_tensor(x) for x in (k_fb, k_ff, q_0, k_fb_init))
    if staged:
        staged = np.size(k_ff) * (n_s * n_s + n_s + 1) <= B.STAGING_MAX_DOUBLES * max(n_u, 1)
    if staged:
        # NumPy in / out: every argument through one pinned block, every result back through one (B.Staging)
        f64 = lambda x: None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        np_p0, np_kff = f64(p_0), f64(k_ff)
        T = np_p0.shape[0]
        if np_kff.ndim != 3 or np_kff.shape[0] != T or np_kff.shape[2] != n_u:
            raise ValueError("k_ff must be (T, H, {})".format(n_u))
        H = np_kff.shape[1]
        np_kfb = f64(k_fb).reshape(T, H - 1, n_u, n_s) if H > 1 else None
        np_q0 = f64(q_0).reshape(T, n_s, n_s) if q_0 is not None else None
        np_kfb0 = f64(k_fb_init).reshape(T, n_u, n_s) if q_0 is not None else None
        st = getattr(hd, "_staging", None)
        if st is None:
            st = hd._staging = B.Staging(dev)
        (p0, kff, kfb, q0, kfb0), (p_all, q_all, bad_slot) = st.stage(
            [np_p0.reshape(T, n_s), np_kff, np_kfb, np_q0, np_kfb0], [(T, H, n_s), (T, H, n_s, n_s), (1,)], zero_copy=True)
    else:
        p0 = B.as_dev(p_0, dev)
        T = p0.shape[0]
        kff = B.as_dev(k_ff, dev)
        if kff.dim() != 3 or kff.shape[0] != T or kff.shape[2] != n_u:
            raise ValueError("k_ff must be (T, H, {})".format(n_u))
        H = kff.shape[1]
        kfb = B.as_dev(k_fb, dev, (T, H - 1, n_u, n_s)) if H > 1 else None
        q0 = B.as_dev(q_0, dev, (T, n_s, n_s)) if q_0 is not None else None
        kfb0 = B.as_dev(k_fb_init, dev, (T, n_u, n_s)) if q0 is not None else None
        p_all = B.empty((T, H, n_s), dev)
        q_all = B.empty((T, H, n_s, n_s), dev)
    a, b = _lin_model(a, b, n_s, n_u)
    ta, tb = B.const_dev(a, dev, (n_s, n_s)), B.const_dev(b, dev, (n_s, n_u))
    tlm, tls = B.const_dev(L_mu, dev, (n_s,)), B.const_dev(L_sigm, dev, (n_s,))
    if staged and check_bounds:
        # the violation counter travels back inside the packed result block (its own D2H copy + synchronisation otherwise)
        bad_slot.zero_()
        n_bad = bad_slot.view(torch.int32)
    else:
        n_bad = B.zeros_i32(1, dev) if check_bounds else None
    with _input_transform(gp, t_z_gp):
        check(lib.sr_multistep_reach(hd.h, T, H, B.ptr(p0), B.ptr(q0), B.ptr(kfb0), B.ptr(kff), B.ptr(kfb),
                                     B.ptr(ta), B.ptr(tb), B.ptr(tlm), B.ptr(tls), float(c_safety),
                                     B.ptr(p_all), B.ptr(q_all), B.ptr(n_bad), B.stream_ptr(dev)))
    if as_t:
        _raise_if_bad(n_bad)
        return p_all, q_all
    if staged:
        p_np, q_np, bad_np = st.fetch()               # (synchronises the stream)
        if check_bounds:
            n_viol = int(bad_np.view(np.int32)[0])
            if n_viol > 0:
                raise AssertionError("all elements of u_b need to be greater than zero! "
                                     "({} query state(s) affected)".format(n_viol))
        _raise_if_chain_failed(hd, dev, True)
        return p_np, q_np
    _raise_if_bad(n_bad)
    _raise_if_chain_failed(hd, dev, n_bad is not None)
    return B.to_numpy(p_all), B.to_numpy(q_all)


def multistep_reachability(p_0, gp, k_fb, k_ff, L_mu, L_sigm, q_0=None, c_safety=1., verbose=1,
                           a=None, b=None, k_fb_init=None, t_z_gp=None):
    """Ellipsoidal overapproximation after n actions (gp_reachability.py:159-212).

    p_0 (n_s,1); k_fb (n-1,n_u,n_s); k_ff (n,n_u).  Returns p_new (n_s,1), q_new (n_s,n_s),
    p_all (n,n_s), q_all (n,n_s,n_s).
    """
    k_fb = np.asarray(k_fb, dtype=np.float64)
    k_ff = np.asarray(k_ff, dtype=np.float64)
    n_, n_u, n_s = np.shape(k_fb)
    n = n_ + 1
    if isinstance(gp, SimpleGPModel):
        q0 = None if q_0 is None else np.asarray(q_0, dtype=np.float64)[None]
        kfb0 = None if (k_fb_init is None or q_0 is None) else np.asarray(k_fb_init, dtype=np.float64)[None]
        p_all, q_all = multistep_reachability_batch(np.asarray(p_0, dtype=np.float64).reshape(1, n_s), gp,
                                                    k_fb[None], k_ff[None], L_mu, L_sigm, q0, c_safety,
                                                    a, b, kfb0, check_bounds=True, t_z_gp=t_z_gp)
        p_all, q_all = p_all[0], q_all[0]
    else:
        p_all = np.empty((n, n_s))
        q_all = np.empty((n, n_s, n_s))
        p_new, q_new = onestep_reachability(p_0, gp, k_ff[0, :, None], L_mu, L_sigm, q_0, k_fb_init,
                                            c_safety, verbose, a, b, t_z_gp)
        p_all[0], q_all[0] = p_new.T, q_new
        for i in range(1, n):
            p_new, q_new = onestep_reachability(p_new, gp, k_ff[i, :, None], L_mu, L_sigm, q_new,
                                                k_fb[i - 1], c_safety, verbose, a, b, t_z_gp)
            p_all[i], q_all[i] = p_new.T, q_new
    return p_all[-1][:, None].copy(), q_all[-1].copy(), p_all, q_all


def lin_ellipsoid_safety_distance_batch(p_center, q_shape, h_mat, h_vec, c_safety=1.0, device=None):
    """d (T,m) = h_mat p + c sqrt(diag(h_mat Q h_mat^T)) - h_vec for T ellipsoids."""
    as_t = B.is_tensor(p_center)
    dev = B.resolve_device(p_center.device if as_t else device)
    p = B.as_dev(p_center, dev)
    T, n_s = p.shape
    q = B.as_dev(q_shape, dev, (T, n_s, n_s))
    hm = B.as_dev(h_mat, dev)
    m = hm.shape[0]
    if hm.dim() != 2 or hm.shape[1] != n_s:
        raise ValueError("h_mat must be (m, {})".format(n_s))
    hv = B.as_dev(np.reshape(B.to_numpy(h_vec) if B.is_tensor(h_vec) else h_vec, (-1,)), dev, (m,))
    d = B.empty((T, m), dev)
    check(lib.sr_safety_distance(dev.index, T, n_s, m, B.ptr(p), B.ptr(q), B.ptr(hm), B.ptr(hv),
                                 float(c_safety), B.ptr(d), B.stream_ptr(dev)))
    return d if as_t else B.to_numpy(d)


def lin_ellipsoid_safety_distance(p_center, q_shape, h_mat, h_vec, c_safety=1.0):
    """Distance between ellipsoid E(p,Q) and polytope h_mat x <= h_vec (gp_reachability.py:215-250).
    d < 0 (element-wise) <=> the ellipsoid is inside the polytope.  Returns (m,1)."""
    m, n_s = np.shape(h_mat)
    assert np.shape(p_center) == (n_s, 1), "p_center has to have shape n_s x 1"
    assert np.shape(q_shape) == (n_s, n_s), "q_shape has to have shape n_s x n_s"
    assert np.shape(h_vec) == (m, 1), "q_shape has to have shape m x 1"
    d = lin_ellipsoid_safety_distance_batch(np.asarray(p_center, dtype=np.float64).reshape(1, n_s),
                                            np.asarray(q_shape, dtype=np.float64)[None], h_mat, h_vec,
                                            c_safety)
    return d.reshape(m, 1)


# ------------------------------------------------------------------------------------------------
# closed-loop roll-outs of the TRUE system against the predicted ellipsoids (gp_reachability.py:253-356).
# `env` is the caller's environment object (duck-typed: ``simulate_onestep(state, action) -> (state, ...)``,
# attributes n_s / n_u); these are host-side bookkeeping around the batched containment kernels.
# ------------------------------------------------------------------------------------------------
def simulate_trajectory(env, p_0, k_fb, k_ff, p_ctrl):
    """Roll the environment forward under u_0 = k_ff[0], u_i = k_fb[i-1] (x_i - p_ctrl[i-1]) + k_ff[i]
    (gp_reachability.py:253-283).  k_ff (n, n_u); k_fb (n-1, n_u*n_s); p_ctrl (n-1, n_s) -> x_all (n+1, n_s)."""
    from .utils import feedback_ctrl
    k_ff = np.asarray(k_ff, dtype=np.float64)
    n, n_u = k_ff.shape
    x = np.asarray(p_0, dtype=np.float64).reshape(-1)
    n_s = x.size
    x_all = np.empty((n + 1, n_s))
    x_all[0] = x
    for i in range(n):
        if i == 0:
            action = k_ff[0]
        else:
            action = feedback_ctrl(x[:, None], k_ff[i, :, None], np.reshape(k_fb[i - 1], (n_u, n_s)),
                                   np.asarray(p_ctrl[i - 1], dtype=np.float64)[:, None])
        x = np.asarray(env.simulate_onestep(x, action)[0], dtype=np.float64).reshape(-1)
        x_all[i + 1] = x
    return x_all


def verify_trajectory_safety(env, p_0, k_fb, k_ff, p_ctrl, h_mat_safe, h_safe, h_mat_obs=None, h_obs=None):
    """True trajectory inside the obstacle-free polytope at steps 1..n-1 and inside the terminal safe polytope at
    the last step (gp_reachability.py:286-320).  Returns (bool, x_all)."""
    from .utils import sample_inside_polytope
    n = np.shape(k_ff)[0]
    x_all = simulate_trajectory(env, p_0, k_fb, k_ff, p_ctrl)
    inside = True
    if h_mat_obs is not None and n > 1:
        inside = bool(np.all(sample_inside_polytope(x_all[1:n], h_mat_obs, h_obs)))
    inside = inside and bool(np.all(sample_inside_polytope(x_all[None, -1, :], h_mat_safe, h_safe)))
    return inside, x_all


def trajectory_inside_ellipsoid(env, p_0, p_all, q_all, k_fb, k_ff):
    """Is the true state at step i inside the predicted ellipsoid (p_all[i], q_all[i])?  (gp_reachability.py:323-356)
    p_all (n, n_s); q_all (n, n_s*n_s) -> bool (n,)."""
    n = np.shape(k_ff)[0]
    n_s = env.n_s
    x_all = simulate_trajectory(env, p_0, k_fb, k_ff, p_all)[1:]
    centred = x_all - np.asarray(p_all, dtype=np.float64).reshape(n, n_s)
    q_all = np.asarray(q_all, dtype=np.float64).reshape(n, n_s, n_s)
    return np.einsum('ij,ij->i', centred, np.linalg.solve(q_all, centred[:, :, None])[:, :, 0]) < 1.0
