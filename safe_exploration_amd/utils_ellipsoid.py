"""Small ellipsoid helpers with the names and semantics of
/root/reference/safe_exploration/utils_ellipsoid.py (sample_inside_ellipsoid :16-33,
distance_to_center :36-60, sum_two_ellipsoids :63-94, sum_ellipsoids :97-151,
ellipsoid_from_rectangle :197-233).

These are O(n_s^3) host conveniences for single ellipsoids.  The batched hot path never
calls them: the same algebra (ellipsoid_from_rectangle + two trace-optimal sums per query) is fused
into sr_ellipsoid_kernel on the device.
"""
import warnings

import numpy as np


def distance_to_center(samples, p_center, q_shape):
    """d_i = (s_i - p)^T Q^-1 (s_i - p) for samples (k,n_s); returns (k,)."""
    centered = np.asarray(samples, dtype=np.float64) - np.reshape(p_center, (1, -1))
    return np.einsum('ij,ji->i', centered, np.linalg.solve(q_shape, centered.T))


def sample_inside_ellipsoid(samples, p_center, q_shape, c=1.):
    """d_i < c per sample."""
    return distance_to_center(samples, p_center, q_shape) < c


def distance_to_center_batch(samples, p_center, q_shape, device=None):
    """Device batch: T ellipsoids (p (T,n_s), q (T,n_s,n_s)) x K samples -> d (T,K).
    samples (K,n_s) are shared by all ellipsoids, samples (T,K,n_s) are per ellipsoid
    (Monte-Carlo verification of a whole batch of trajectories, sampling_models.py:33-107)."""
    from . import _buffers as B
    from ._lib import lib, check
    as_t = B.is_tensor(p_center)
    dev = B.resolve_device(p_center.device if as_t else device)
    p = B.as_dev(p_center, dev)
    T, n_s = p.shape
    q = B.as_dev(q_shape, dev, (T, n_s, n_s))
    smp = B.as_dev(samples, dev)
    per_t = 1 if smp.dim() == 3 else 0
    K = smp.shape[-2]
    if smp.shape[-1] != n_s or (per_t and smp.shape[0] != T):
        raise ValueError("samples must be (K, n_s) or (T, K, n_s)")
    d = B.empty((T, K), dev)
    check(lib.sr_distance_to_center(dev.index, T, K, n_s, B.ptr(smp), per_t, B.ptr(p), B.ptr(q), B.ptr(d),
                                    B.stream_ptr(dev)))
    return d if as_t else B.to_numpy(d)


def sample_inside_ellipsoid_batch(samples, p_center, q_shape, c=1., device=None):
    """d < c for T ellipsoids x K samples (see distance_to_center_batch)."""
    return distance_to_center_batch(samples, p_center, q_shape, device) < c


def sum_two_ellipsoids(p_1, q_1, p_2, q_2, c=None):
    """Outer ellipsoid of the Minkowski sum; c = sqrt(tr q_1 / tr q_2) minimises the trace."""
    if c is None:
        c = np.sqrt(np.trace(q_1) / np.trace(q_2))
    return p_1 + p_2, (1.0 + 1.0 / c) * q_1 + (1.0 + c) * q_2


def sum_ellipsoids(p, q, l=None):
    """Outer ellipsoid of the sum of n >= 2 ellipsoids, tight along direction l.
    p (n,m), q (n,m,m), l (m,1)."""
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    if l is None:
        l = np.diag(q[0])
        warnings.warn("Bad heuristic for choice of l. Might have to think of better ones")
    n = p.shape[0]
    assert n >= 2, "Need at least two input ellipsoids"
    if n == 2:
        return sum_two_ellipsoids(p[0, :, None], q[0], p[1, :, None], q[1])
    l = np.asarray(l, dtype=np.float64)
    c_i = np.sqrt(np.array([np.dot(l.T, np.dot(q[i], l)) for i in range(n)]).reshape(n))
    q_new = np.sum(c_i) * np.einsum('i,ijk->jk', 1.0 / c_i, q)
    return np.sum(p, axis=0)[:, None], q_new


def ellipsoid_from_rectangle(u_b):
    """Minimum-trace axis-aligned ellipsoid covering the box [-u_b, u_b]: diag(n u_b^2)."""
    u_b = np.asarray(u_b)
    assert u_b.ndim == 1, "lb and ub need to be 1-dimensional (1darrays)!"
    assert np.all(u_b > 0), "all elements of u_b need to be greater than zero!"
    return np.diag(len(u_b) * u_b ** 2)
