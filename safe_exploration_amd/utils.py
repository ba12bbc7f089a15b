"""Helpers of the reachability path (counterparts of the four hot-path functions of
/root/reference/safe_exploration/utils.py: compute_remainder_overapproximations :108-144,
sample_inside_polytope :38-56, feedback_ctrl :59-65, array_of_vec_to_array_of_mat :208-227)."""
import numpy as np

from . import _buffers as B
from ._lib import lib, check


def compute_remainder_overapproximations_batch(q, k_fb, l_mu, l_sigma, device=None):
    """q (T,n_s,n_s), k_fb (T,n_u,n_s) -> u_mu (T,n_s), u_sigma (T,n_s) (HIP, one thread per query)."""
    as_t = B.is_tensor(q)
    dev = B.resolve_device(q.device if as_t else device)
    tq = B.as_dev(q, dev)
    T, n_s, _ = tq.shape
    tk = B.as_dev(k_fb, dev)
    n_u = tk.shape[1]
    if tuple(tk.shape) != (T, n_u, n_s):
        raise ValueError("k_fb must be (T, n_u, n_s)")
    tlm, tls = B.as_dev(np.reshape(l_mu, (-1,)), dev, (n_s,)), B.as_dev(np.reshape(l_sigma, (-1,)), dev, (n_s,))
    u_mu, u_sigma = B.empty((T, n_s), dev), B.empty((T, n_s), dev)
    check(lib.sr_remainder_overapprox(dev.index, T, n_s, n_u, B.ptr(tq), B.ptr(tk), B.ptr(tlm),
                                      B.ptr(tls), B.ptr(u_mu), B.ptr(u_sigma), B.stream_ptr(dev)))
    return (u_mu, u_sigma) if as_t else (B.to_numpy(u_mu), B.to_numpy(u_sigma))


def compute_remainder_overapproximations(q, k_fb, l_mu, l_sigma):
    """Hyper-rectangles over-approximating the Lagrange remainders of mean and std-dev
    (utils.py:108-144): r^2 = lambda_max(Q (I + K^T K)); u_mu = l_mu r^2; u_sigma = l_sigma r.
    q (n_s,n_s) symmetric, k_fb (n_u,n_s).  Returns real 1-d arrays of length n_s (the reference
    returns complex dtype with zero imaginary part because it goes through scipy.linalg.eig)."""
    u_mu, u_sigma = compute_remainder_overapproximations_batch(
        np.asarray(q, dtype=np.float64)[None], np.asarray(k_fb, dtype=np.float64)[None], l_mu, l_sigma)
    return u_mu[0], u_sigma[0]


def sample_inside_polytope(x, a, b):
    """For k samples x (k,n): all(a x_i - b < 0) per sample (utils.py:38-56).  Host predicate."""
    x = np.asarray(x)
    c = np.dot(a, x.T) - np.reshape(b, (-1, 1))
    return np.all(c < 0, axis=0).squeeze()


def feedback_ctrl(x, k_ff, k_fb=None, p=None):
    """u = K (x - p) + k (utils.py:59-65)."""
    if k_fb is None:
        return k_ff
    return np.dot(k_fb, (x - p)) + k_ff


def dlqr(a, b, q, r):
    """Infinite-horizon discrete LQR for x+ = a x + b u, u = -k x (utils.py:20-35): returns the gain k (n_u, n_s),
    the Riccati solution x and the closed-loop eigenvalues of a - b k.  Host helper of the callers that build the
    feedback gains k_fb handed to the reachability functions."""
    import scipy.linalg as sla
    a, b = np.atleast_2d(np.asarray(a, dtype=np.float64)), np.atleast_2d(np.asarray(b, dtype=np.float64))
    q, r = np.atleast_2d(np.asarray(q, dtype=np.float64)), np.atleast_2d(np.asarray(r, dtype=np.float64))
    x = sla.solve_discrete_are(a, b, q, r)
    k = np.linalg.solve(b.T.dot(x).dot(b) + r, b.T.dot(x).dot(a))
    return k, x, np.linalg.eigvals(a - b.dot(k))


def array_of_vec_to_array_of_mat(array_of_vec, n, m):
    """(T, n*m) -> (T, n, m) (utils.py:208-227)."""
    return np.reshape(array_of_vec, (-1, n, m))


def reshape_derivatives_3d_to_2d(derivative_3d):
    """(r, s, n_in) derivative tensor -> (r*s, n_in), the layout CasADi callbacks need (utils.py:357-380)."""
    r, s, n_in = np.shape(derivative_3d)
    return np.reshape(derivative_3d, (r * s, n_in))


def print_ellipsoid(p_center, q_shape, text="ellipsoid", visualize=False):
    print("\n")
    print("===== {} =====".format(text))
    print("center:")
    print(p_center)
    print("==========")
    print("diagonal of shape matrix:")
    print(np.diag(q_shape))
    print("===============")
