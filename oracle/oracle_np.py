"""NumPy/SciPy restatement of the reference's GP-inference + ellipsoid-reachability path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Pinning status
--------------
* Ellipsoid / reachability algebra (``onestep_reachability`` ...): PINNED against the
  reference's own functions, imported in the build container from /root/reference
  (``tests/golden/make_golden.py``); fixtures in ``tests/golden/reach_*.npz``.
* RBF kernel + posterior formulas (``rbf_kernel``, ``gp_predict``): checked against the
  reference's in-tree restatement ``gp_models_utils_casadi._k_rbf/_unscaled_dist/gp_pred``
  executed on numbers (casadi's 6 array functions replaced by numpy equivalents in the
  generator script only); fixtures ``tests/golden/gp_*.npz``.
* Matern-5/2 / lin_* kernels, Gaussian moment propagation, Monte-Carlo particle propagation: PINNED against the
  reference's in-tree formulas / functions evaluated on numbers (``kern_*.npz``, ``moments_*.npz``, ``mc_*.npz``).
* Second-order outputs of the non-RBF kernels (``gp_linearize_extras_k``), marginal likelihood and its gradient
  (``gp_nll_grad``), greedy max-variance selection (``choose_datapoints_maxvar``): the reference delegates these
  to CasADi's AD resp. GPy's optimiser / model objects, so there are no reference numbers; they are checked
  against central differences, scikit-learn, and refits from scratch in tests/test_oracle_golden.py.
* GPy boundary (``GPRegression`` internals: +1e-8 jitter, 1e-15 variance clip, r2>=0 clip):
  **parity unpinned** -- GPy (unpinned in the reference's setup.py:20) is not installed and
  no stored GPy outputs exist in the reference tree.  Those three details are restated from
  knowledge of GPy and flagged below.

Every function cites the reference file:line it follows (paths relative to
/root/reference/safe_exploration/).
"""
import numpy as np
import scipy.linalg as sla

GPY_JITTER = 1e-8        # GPy exact_gaussian_inference adds 1e-8 to diag(K) [GPy, unverifiable here]
GPY_VAR_CLIP = 1e-15     # GPy GP._raw_predict clips var to >=1e-15       [GPy, unverifiable here]


# --------------------------------------------------------------------------- kernel
def unscaled_dist_sq(x, y):
    """r^2 = -2 x y^T + |x|^2 + |y|^2, clipped at 0.

    ssm_gpy/gp_models_utils_casadi.py:160-174 (mirrors GPy stationary._unscaled_dist,
    which additionally clips r2 at 0 before the sqrt).
    """
    x1sq = np.sum(x ** 2, axis=1)
    x2sq = np.sum(y ** 2, axis=1)
    r2 = -2.0 * x.dot(y.T) + x1sq[:, None] + x2sq[None, :]
    return np.clip(r2, 0.0, np.inf)


def rbf_kernel(x, y, variance, lengthscale):
    """ARD RBF k(x,y) = variance * exp(-0.5 r^2), r = dist(x/l, y/l).

    ssm_gpy/gp_models_utils_casadi.py:17-40.
    """
    ls = np.asarray(lengthscale, dtype=np.float64).reshape(1, -1)
    r2 = unscaled_dist_sq(x / ls, y / ls)
    return variance * np.exp(-0.5 * r2)


# --------------------------------------------------------------------------- fit
def gp_fit(Z, Y, lengthscale, signal_var, noise_var):
    """Cache what SimpleGPModel.train(opt_hyp=False) caches: beta and inv_K per output.

    ssm_gpy/gaussian_process.py:231-275: per output d a GPRegression(Z, y_d, kern_d); noise
    fixed (noise_var here ALREADY includes the +1e-5 ``noise_diag`` of :189,252-253);
    inv_K[d] = posterior.woodbury_inv = (K + (noise+1e-8) I)^-1, beta[:,d] = woodbury_vector.

    Parameters: Z (N,D); Y (N,n_out); lengthscale (n_out,D); signal_var (n_out,); noise_var (n_out,)
    Returns: beta (N,n_out), inv_K list of (N,N), chol list of lower factors L (K_y = L L^T)
    """
    Z = np.asarray(Z, np.float64)
    Y = np.asarray(Y, np.float64)
    N, n_out = Y.shape
    beta = np.empty((N, n_out))
    inv_K, chol = [], []
    for d in range(n_out):
        K = rbf_kernel(Z, Z, signal_var[d], lengthscale[d])
        Ky = K + (noise_var[d] + GPY_JITTER) * np.eye(N)
        L = sla.cholesky(Ky, lower=True)
        Li = sla.solve_triangular(L, np.eye(N), lower=True)
        Ki = Li.T.dot(Li)          # GPy pdinv: dpotri on the Cholesky factor
        inv_K.append(Ki)
        chol.append(L)
        beta[:, d] = sla.cho_solve((L, True), Y[:, d])
    return beta, inv_K, chol


# --------------------------------------------------------------------------- predict
def gp_predict(x_new, Z, beta, inv_K, lengthscale, signal_var, compute_gradients=True):
    """Batched posterior mean / variance / mean-Jacobian, EXPLICIT-INVERSE route.

    mean, var: ssm_gpy/gp_models_utils_casadi.py:177-197 (gp_pred):
        mu = k* beta ; sigma2 = k(x,x) - sum((k* K^-1) o k*, axis=1)
    batched shapes: ssm_gpy/gaussian_process.py:546-568 (predict) -> (T,n_out),(T,n_out)
    jacobian: ssm_gpy/gaussian_process.py:570-596 (predictive_gradients) -> (T,n_out,D);
        RBF closed form d mu/dx = sum_i beta_i k*_i (z_i - x)/l^2
        (what CasADi AD of gp_pred yields, gp_models_utils_casadi.py:272-280).
    Variance clipped at 1e-15 like GPy's _raw_predict [unverifiable here].
    """
    x_new = np.asarray(x_new, np.float64)
    T, D = x_new.shape
    n_out = beta.shape[1]
    mu = np.empty((T, n_out))
    var = np.empty((T, n_out))
    jac = np.empty((T, n_out, D)) if compute_gradients else None
    for d in range(n_out):
        ks = rbf_kernel(x_new, Z, signal_var[d], lengthscale[d])      # (T,N)
        mu[:, d] = ks.dot(beta[:, d])
        var[:, d] = signal_var[d] - np.sum(ks.dot(inv_K[d]) * ks, axis=1)
        if compute_gradients:
            w = ks * beta[:, d][None, :]                              # (T,N)
            l2 = np.asarray(lengthscale[d], np.float64) ** 2
            # sum_i w_ti (z_ij - x_tj) / l_j^2
            jac[:, d, :] = (w.dot(Z) - w.sum(axis=1)[:, None] * x_new) / l2[None, :]
    var = np.clip(var, GPY_VAR_CLIP, np.inf)
    if compute_gradients:
        return mu, var, jac
    return mu, var


def gp_predict_chol(x_new, Z, beta, chol, lengthscale, signal_var):
    """Second algebraic route (triangular solve against the cached factor):
    sigma2 = k(x,x) - |L^-1 k*|^2.  Used to cross-check gp_predict."""
    x_new = np.asarray(x_new, np.float64)
    T = x_new.shape[0]
    n_out = beta.shape[1]
    mu = np.empty((T, n_out))
    var = np.empty((T, n_out))
    for d in range(n_out):
        ks = rbf_kernel(x_new, Z, signal_var[d], lengthscale[d])
        mu[:, d] = ks.dot(beta[:, d])
        v = sla.solve_triangular(chol[d], ks.T, lower=True)
        var[:, d] = signal_var[d] - np.sum(v * v, axis=0)
    return mu, np.clip(var, GPY_VAR_CLIP, np.inf)


def gp_linearize_extras(x, Z, beta, inv_K, lengthscale, signal_var):
    """Single-query d sigma2/dx and Hessian of mu (contract
    state_space_models.py:106-138).  RBF closed forms:
        d sigma2/dx = -2 sum_i (K^-1 k*)_i k*_i (z_i - x)/l^2
        d2 mu/dx2   = sum_i beta_i k*_i [ (z_i-x)(z_i-x)^T/(l^2 l^2^T) - diag(l^-2) ]
    Returns jac_var (n_out,D), hess_mu (n_out,D,D)."""
    x = np.asarray(x, np.float64).reshape(1, -1)
    D = x.shape[1]
    n_out = beta.shape[1]
    jv = np.empty((n_out, D))
    hm = np.empty((n_out, D, D))
    for d in range(n_out):
        l2 = np.asarray(lengthscale[d], np.float64) ** 2
        ks = rbf_kernel(x, Z, signal_var[d], lengthscale[d])[0]
        diff = (Z - x) / l2[None, :]                                  # (N,D)
        g = inv_K[d].dot(ks)
        jv[d] = -2.0 * (g * ks).dot(diff)
        w = beta[:, d] * ks
        hm[d] = np.einsum('i,ij,ik->jk', w, diff, diff) - np.diag(w.sum() / l2)
    return jv, hm


# --------------------------------------------------------------------------- ellipsoids
def ellipsoid_from_rectangle(u_b):
    """utils_ellipsoid.py:197-233: q = diag(n * u_b^2); asserts u_b > 0."""
    u_b = np.asarray(u_b)
    assert u_b.ndim == 1
    assert np.all(u_b > 0)
    return np.diag(len(u_b) * u_b ** 2)


def sum_two_ellipsoids(p_1, q_1, p_2, q_2, c=None):
    """utils_ellipsoid.py:63-94 (trace-optimal c when None)."""
    if c is None:
        c = np.sqrt(np.trace(q_1) / np.trace(q_2))
    return p_1 + p_2, (1 + (1. / c)) * q_1 + (1 + c) * q_2


def compute_remainder_overapproximations(q, k_fb, l_mu, l_sigma):
    """utils.py:108-144: r^2 = max eig(Q (I + K^T K)); u_mu = l_mu r^2; u_sigma = l_sigma r.
    The reference takes scipy.linalg.eig (complex dtype, zero imaginary part); here the real
    part is returned."""
    n_u, n_s = np.shape(k_fb)
    s = np.hstack((np.eye(n_s), k_fb.T))
    b = s.dot(s.T)
    evals = sla.eigvals(q.dot(b))
    r_sqr = np.max(evals.real)
    return l_mu * r_sqr, l_sigma * np.sqrt(r_sqr)


def lin_ellipsoid_safety_distance(p_center, q_shape, h_mat, h_vec, c_safety=1.0):
    """gp_reachability.py:215-250."""
    d_center = h_mat.dot(p_center)
    d_shape = c_safety * np.sqrt(np.sum(q_shape.dot(h_mat.T) * h_mat.T, axis=0)[:, None])
    return d_center + d_shape - h_vec


def distance_to_center(samples, p_center, q_shape):
    """utils_ellipsoid.py:36-60."""
    pc = samples - p_center.T
    return np.sum(pc * sla.solve(q_shape, pc.T).T, axis=1)


def sample_inside_polytope(x, a, b):
    """utils.py:38-56."""
    c = a.dot(x.T) - b.reshape(-1, 1)
    return np.all(c < 0, axis=0)


# --------------------------------------------------------------------------- reachability
def onestep_reachability_from_gp(p, q, k_ff, k_fb, mu, var, jac, l_mu, l_sigma, c_safety, a, b):
    """One query of gp_reachability.py:19-156 given the GP outputs (mu,var (n_s,), jac (n_s,D)).

    p (n_s,), q (n_s,n_s) or None, k_ff (n_u,), k_fb (n_u,n_s) or None.
    Returns p1 (n_s,), q1 (n_s,n_s) real."""
    n_s = p.shape[0]
    if q is None:                                   # gp_reachability.py:65-88
        q1 = ellipsoid_from_rectangle(c_safety * np.sqrt(var))
        p1 = a.dot(p) + b.dot(k_ff) + mu
        return p1, q1
    a_mu, b_mu = jac[:, :n_s], jac[:, n_s:]         # :110-111
    H = a + a_mu + (b_mu + b).dot(k_fb)             # :114
    p0 = mu + a.dot(p) + b.dot(k_ff)                # :115
    Q0 = H.dot(q).dot(H.T)                          # :117
    ub_mean, ub_sigma = compute_remainder_overapproximations(q, k_fb, l_mu, l_sigma)  # :125
    b_sigma_eps = c_safety * (np.sqrt(var) + ub_sigma)                                 # :127
    Qs = ellipsoid_from_rectangle(b_sigma_eps)      # :129
    Qm = ellipsoid_from_rectangle(ub_mean)          # :136
    _, QL = sum_two_ellipsoids(0, Qs, 0, Qm)        # :143
    p1, q1 = sum_two_ellipsoids(0, QL, p0, Q0)      # :148
    return p1, q1


def _predict_one(model, z):
    mu, var, jac = gp_predict(z[None, :], model["Z"], model["beta"], model["inv_K"],
                              model["lengthscale"], model["signal_var"], True)
    return mu[0], var[0], jac[0]


def onestep_reachability_batch(model, p, q, k_ff, k_fb, l_mu, l_sigma, c_safety=1.0, a=None, b=None):
    """Batched (leading T axis) one-step reachability, vectorised GP + per-query ellipsoid algebra.

    model: dict(Z, beta, inv_K, lengthscale, signal_var).  p (T,n_s); q (T,n_s,n_s) or None;
    k_ff (T,n_u); k_fb (T,n_u,n_s) or None.  Returns p1 (T,n_s), q1 (T,n_s,n_s), var (T,n_s)."""
    T, n_s = p.shape
    n_u = k_ff.shape[1]
    if a is None:
        a, b = np.eye(n_s), np.zeros((n_s, n_u))
    x = np.hstack((p, k_ff))
    mu, var, jac = gp_predict(x, model["Z"], model["beta"], model["inv_K"],
                              model["lengthscale"], model["signal_var"], True)
    p1 = np.empty((T, n_s))
    q1 = np.empty((T, n_s, n_s))
    for t in range(T):
        p1[t], q1[t] = onestep_reachability_from_gp(
            p[t], None if q is None else q[t], k_ff[t], None if k_fb is None else k_fb[t],
            mu[t], var[t], jac[t], l_mu, l_sigma, c_safety, a, b)
    return p1, q1, var


def multistep_reachability_batch(model, p0, k_fb, k_ff, l_mu, l_sigma, q0=None, c_safety=1.0,
                                 a=None, b=None, k_fb_init=None):
    """gp_reachability.py:159-212 with a leading T axis.
    k_fb (T,H-1,n_u,n_s), k_ff (T,H,n_u), k_fb_init (T,n_u,n_s) or None (needed iff q0 given).
    Returns p_all (T,H,n_s), q_all (T,H,n_s,n_s)."""
    T, H, n_u = k_ff.shape
    n_s = p0.shape[1]
    p_all = np.empty((T, H, n_s))
    q_all = np.empty((T, H, n_s, n_s))
    p, q, _ = onestep_reachability_batch(model, p0, q0, k_ff[:, 0], k_fb_init, l_mu, l_sigma,
                                         c_safety, a, b)
    p_all[:, 0], q_all[:, 0] = p, q
    for i in range(1, H):
        p, q, _ = onestep_reachability_batch(model, p, q, k_ff[:, i], k_fb[:, i - 1], l_mu, l_sigma,
                                             c_safety, a, b)
        p_all[:, i], q_all[:, i] = p, q
    return p_all, q_all


# --------------------------------------------------------------------------- vectorised CPU baseline
def onestep_reachability_vectorised(model, p, q, k_ff, k_fb, l_mu, l_sigma, c_safety, a, b):
    """'B-vec' CPU baseline: same algebra as onestep_reachability_batch's ellipsoid
    branch but with the per-query algebra vectorised over T (np.linalg.eigvals batched)."""
    T, n_s = p.shape
    x = np.hstack((p, k_ff))
    mu, var, jac = gp_predict(x, model["Z"], model["beta"], model["inv_K"],
                              model["lengthscale"], model["signal_var"], True)
    a_mu, b_mu = jac[:, :, :n_s], jac[:, :, n_s:]
    H = a[None] + a_mu + np.einsum('tiu,tuj->tij', b_mu + b[None], k_fb)
    p0 = mu + p.dot(a.T) + k_ff.dot(b.T)
    Q0 = np.einsum('tij,tjk,tlk->til', H, q, H)
    B = np.eye(n_s)[None] + np.einsum('tui,tuj->tij', k_fb, k_fb)
    r2 = np.max(np.linalg.eigvals(np.einsum('tij,tjk->tik', q, B)).real, axis=1)
    ub_mean = l_mu[None] * r2[:, None]
    ub_sigma = l_sigma[None] * np.sqrt(r2)[:, None]
    ds = n_s * (c_safety * (np.sqrt(var) + ub_sigma)) ** 2
    dm = n_s * ub_mean ** 2
    c1 = np.sqrt(ds.sum(1) / dm.sum(1))
    dL = (1 + 1 / c1)[:, None] * ds + (1 + c1)[:, None] * dm
    c2 = np.sqrt(dL.sum(1) / np.trace(Q0, axis1=1, axis2=2))
    q1 = (1 + c2)[:, None, None] * Q0
    idx = np.arange(n_s)
    q1[:, idx, idx] += (1 + 1 / c2)[:, None] * dL
    return p0, q1, var


# --------------------------------------------------------------------------- synthetic configs
def make_synthetic(seed, N, n_s, n_u, T, noise=1e-2, sf2=1.0):
    """Seeded synthetic problem of SURVEY.md 8(d): Z~U[-1,1], y_d = sf*(sin(2 z.w_d)+0.05 randn),
    l~U[0.5,1.5], signal variance sf2 (1 in the survey), sn2 = noise*sf2 (+1e-5 noise_diag);
    queries p~0.3 randn, k_ff~0.1 randn, k_fb~0.1 randn, Q = 0.01 A A^T + 0.01 I."""
    rng = np.random.default_rng(seed)
    D = n_s + n_u
    Z = rng.uniform(-1, 1, (N, D))
    Wd = rng.standard_normal((n_s, D))
    Y = np.sqrt(sf2) * (np.sin(2.0 * Z.dot(Wd.T)) + 0.05 * rng.standard_normal((N, n_s)))
    ls = rng.uniform(0.5, 1.5, (n_s, D))
    sn2 = np.full(n_s, noise * sf2 + 1e-5)
    sf2 = np.full(n_s, float(sf2))
    p = 0.3 * rng.standard_normal((T, n_s))
    k_ff = 0.1 * rng.standard_normal((T, n_u))
    k_fb = 0.1 * rng.standard_normal((T, n_u, n_s))
    A = rng.standard_normal((T, n_s, n_s))
    Q = 0.01 * np.einsum('tij,tkj->tik', A, A) + 0.01 * np.eye(n_s)[None]
    return dict(Z=Z, Y=Y, lengthscale=ls, signal_var=sf2, noise_var=sn2,
                p=p, k_ff=k_ff, k_fb=k_fb, Q=Q)


# --------------------------------------------------------------------------- non-RBF kernel types
SQRT5 = np.sqrt(5.0)


def mat52_kernel(x, y, variance, lengthscale):
    """ssm_gpy/gp_models_utils_casadi.py:43-69: variance (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)."""
    ls = np.asarray(lengthscale, dtype=np.float64).reshape(-1) * np.ones(x.shape[1])
    r = np.sqrt(unscaled_dist_sq(x / ls[None, :], y / ls[None, :]))
    return variance * (1.0 + SQRT5 * r + 5.0 / 3.0 * r ** 2) * np.exp(-SQRT5 * r)


def lin_kernel(x, y, variances):
    """ssm_gpy/gp_models_utils_casadi.py:131-157: (x sqrt(v)) (y sqrt(v))^T."""
    v = np.asarray(variances, dtype=np.float64).reshape(-1) * np.ones(x.shape[1])
    return (x * np.sqrt(v)[None, :]).dot((y * np.sqrt(v)[None, :]).T)


def kernel_matrix(kern_type, hyp, x, y):
    """K(x, y) for the reference's kernel identifiers (gp_models_utils_casadi.py:200-231).  The
    lin_* types act with their product part on input dimension 1 only (:83-84, :113-114)."""
    if kern_type == "rbf":
        return rbf_kernel(x, y, hyp["variance"], hyp["lengthscale"])
    if kern_type == "mat52":
        return mat52_kernel(x, y, hyp["variance"], hyp["lengthscale"])
    x1, y1 = x[:, 1:2], y[:, 1:2]
    k_lin = lin_kernel(x, y, hyp["linear.variances"])
    k_prod_lin = lin_kernel(x1, y1, hyp["prod.linear.variances"])
    if kern_type == "lin_rbf":
        return k_prod_lin * rbf_kernel(x1, y1, hyp["prod.rbf.variance"], hyp["prod.rbf.lengthscale"]) + k_lin
    if kern_type == "lin_mat52":
        return k_prod_lin * mat52_kernel(x1, y1, hyp["prod.mat52.variance"], hyp["prod.mat52.lengthscale"]) + k_lin
    raise ValueError("Unknown kernel {}".format(kern_type))


def kernel_diag(kern_type, hyp, x):
    """k(x_t, x_t) (the diag_only branches of gp_models_utils_casadi.py)."""
    if kern_type in ("rbf", "mat52"):
        return np.full(x.shape[0], float(hyp["variance"]))
    vp = np.asarray(hyp["prod.linear.variances"], dtype=np.float64).reshape(-1)[0]
    vl = np.asarray(hyp["linear.variances"], dtype=np.float64).reshape(-1) * np.ones(x.shape[1])
    var = hyp["prod.rbf.variance"] if kern_type == "lin_rbf" else hyp["prod.mat52.variance"]
    return vp * x[:, 1] ** 2 * float(var) + (x ** 2).dot(vl)


def gp_fit_k(Z, Y, kern_types, hyp, noise_var):
    """gp_fit for arbitrary kernel identifiers (one per output)."""
    N, n_out = Y.shape
    beta = np.empty((N, n_out))
    inv_K = []
    for d in range(n_out):
        Ky = kernel_matrix(kern_types[d], hyp[d], Z, Z) + (noise_var[d] + GPY_JITTER) * np.eye(N)
        L = sla.cholesky(Ky, lower=True)
        Li = sla.solve_triangular(L, np.eye(N), lower=True)
        inv_K.append(Li.T.dot(Li))
        beta[:, d] = sla.cho_solve((L, True), Y[:, d])
    return beta, inv_K


def gp_predict_k(x_new, Z, beta, inv_K, kern_types, hyp):
    """mean / variance for arbitrary kernel identifiers, formulas of gp_pred
    (gp_models_utils_casadi.py:177-197).  Returns mu (T,n_out), var (T,n_out)."""
    T = x_new.shape[0]
    n_out = beta.shape[1]
    mu = np.empty((T, n_out))
    var = np.empty((T, n_out))
    for d in range(n_out):
        ks = kernel_matrix(kern_types[d], hyp[d], x_new, Z)
        mu[:, d] = ks.dot(beta[:, d])
        var[:, d] = kernel_diag(kern_types[d], hyp[d], x_new) - np.sum(ks.dot(inv_K[d]) * ks, axis=1)
    return mu, np.clip(var, GPY_VAR_CLIP, np.inf)


def gp_mean_jacobian_fd(x_new, Z, beta, kern_types, hyp, eps=1e-6):
    """central-difference d mu/dx (T,n_out,D): an algebra-free check of the analytic Jacobians."""
    T, D = x_new.shape
    n_out = beta.shape[1]
    jac = np.empty((T, n_out, D))
    for j in range(D):
        e = np.zeros(D)
        e[j] = eps
        for d in range(n_out):
            kp = kernel_matrix(kern_types[d], hyp[d], x_new + e, Z)
            km = kernel_matrix(kern_types[d], hyp[d], x_new - e, Z)
            jac[:, d, j] = (kp - km).dot(beta[:, d]) / (2 * eps)
    return jac


def _stationary_radial(kind, r):
    """kappa(r), g = kappa'(r)/r, h = g'(r)/r of the two stationary families (unit variance)."""
    if kind == "rbf":
        k = np.exp(-0.5 * r ** 2)
        return k, -k, k
    e = np.exp(-SQRT5 * r)
    return (1.0 + SQRT5 * r + 5.0 / 3.0 * r ** 2) * e, -(5.0 / 3.0) * (1.0 + SQRT5 * r) * e, (25.0 / 3.0) * e


def kernel_derivatives(kern_type, hyp, x, Z):
    """First and second derivatives of k(x, z_i) w.r.t. the single query x (D,) for the reference's four
    kernel identifiers (formulas differentiated by hand from gp_models_utils_casadi.py:17-157; the
    reference itself leaves this to CasADi's AD).  Returns k (N,), grad (N,D), hess (N,D,D), and the
    gradient of the prior variance k(x,x) (D,)."""
    x = np.asarray(x, np.float64).reshape(-1)
    N, D = Z.shape
    grad = np.zeros((N, D))
    hess = np.zeros((N, D, D))
    if kern_type in ("rbf", "mat52"):
        ls = np.asarray(hyp["lengthscale"], np.float64).reshape(-1) * np.ones(D)
        u = (x[None, :] - Z) / ls[None, :] ** 2                       # (N,D)
        r = np.sqrt(np.sum(((x[None, :] - Z) / ls[None, :]) ** 2, axis=1))
        kap, g, h = _stationary_radial(kern_type, r)
        v = float(hyp["variance"])
        k = v * kap
        grad = v * g[:, None] * u
        hess = v * (h[:, None, None] * u[:, :, None] * u[:, None, :] + g[:, None, None] * np.diag(1.0 / ls ** 2)[None])
        return k, grad, hess, np.zeros(D)
    st = "rbf" if kern_type == "lin_rbf" else "mat52"
    ell = float(np.asarray(hyp["prod.%s.lengthscale" % st]).reshape(-1)[0])
    vs = float(hyp["prod.%s.variance" % st])
    vp = float(np.asarray(hyp["prod.linear.variances"]).reshape(-1)[0])
    vl = np.asarray(hyp["linear.variances"], np.float64).reshape(-1) * np.ones(D)
    x1, z1 = x[1], Z[:, 1]
    u1 = (x1 - z1) / ell ** 2
    kap, g, h = _stationary_radial(st, np.abs(x1 - z1) / ell)
    c = vp * vs * z1
    k = c * x1 * kap + Z.dot(vl * x)
    grad = Z * vl[None, :]
    grad[:, 1] += c * (kap + x1 * g * u1)
    hess[:, 1, 1] = c * (2.0 * g * u1 + x1 * (h * u1 ** 2 + g / ell ** 2))
    gxx = 2.0 * vl * x
    gxx[1] += 2.0 * vp * vs * x1
    return k, grad, hess, gxx


def gp_mean_jacobian_k(x_new, Z, beta, kern_types, hyp):
    """Analytic d mu/dx (T, n_out, D) for arbitrary kernel identifiers: sum_i beta_i d k(x, z_i)/dx with the
    closed-form kernel gradients of ``kernel_derivatives`` (hand-differentiated from
    gp_models_utils_casadi.py:17-157; the reference obtains this Jacobian from CasADi's AD, :272-280).
    Pinned against torch-fp64 autograd of the reference's kernel formulas in tests/test_oracle_golden.py."""
    x_new = np.asarray(x_new, np.float64)
    T, D = x_new.shape
    n_out = beta.shape[1]
    jac = np.empty((T, n_out, D))
    for t in range(T):
        for d in range(n_out):
            jac[t, d] = beta[:, d].dot(kernel_derivatives(kern_types[d], hyp[d], x_new[t], Z)[1])
    return jac


def gp_linearize_extras_k(x, Z, beta, inv_K, kern_types, hyp):
    """d sigma2/dx and Hessian of mu for arbitrary kernel identifiers (the second-order outputs of
    linearize_predict(jacobians=True), state_space_models.py:106-138):
        d sigma2/dx = d k(x,x)/dx - 2 sum_i (K^-1 k*)_i d k*_i/dx ,   d2 mu/dx2 = sum_i beta_i d2 k*_i/dx2."""
    n_out = beta.shape[1]
    D = Z.shape[1]
    jv, hm = np.empty((n_out, D)), np.empty((n_out, D, D))
    for d in range(n_out):
        k, grad, hess, gxx = kernel_derivatives(kern_types[d], hyp[d], x, Z)
        jv[d] = gxx - 2.0 * inv_K[d].dot(k).dot(grad)
        hm[d] = np.einsum('i,ijk->jk', beta[:, d], hess)
    return jv, hm


def make_hyp(kern_type, rng, D):
    """random hyper-parameters with the key names of SimpleGPModel._create_hyp_dict
    (ssm_gpy/gaussian_process.py:491-544)."""
    if kern_type in ("rbf", "mat52"):
        return {"lengthscale": rng.uniform(0.5, 1.5, D), "variance": float(rng.uniform(0.5, 1.5))}
    st = "rbf" if kern_type == "lin_rbf" else "mat52"
    return {"prod.%s.lengthscale" % st: np.array([rng.uniform(0.5, 1.5)]),
            "prod.%s.variance" % st: float(rng.uniform(0.5, 1.5)),
            "prod.linear.variances": np.array([rng.uniform(0.5, 1.5)]),
            "linear.variances": rng.uniform(0.2, 1.0, D)}


# --------------------------------------------------------------------------- Gaussian moment propagation (8(f).4)
def one_step_moments(mu_x, sigma_x, k_ff, k_fb, mu_g, var_g, jac_g, a, b, taylor=True):
    """Literal restatement of uncertainty_propagation_casadi.py:57-87 (Taylor) / :260-283 (mean-equivalent)
    for one query: builds Sigma_z, Sigma_zg, Sigma_all and applies [a b I].  mu_x (n_s,), sigma_x (n_s,n_s) or
    None, k_ff (n_u,), k_fb (n_u,n_s).  Returns mu_new (n_s,), sigma_new (n_s,n_s)."""
    n_s, n_u = mu_x.shape[0], k_ff.shape[0]
    if sigma_x is None:
        return a.dot(mu_x) + b.dot(k_ff) + mu_g, np.diag(var_g)
    sigma_u = k_fb.dot(sigma_x).dot(k_fb.T)
    sigma_xu = sigma_x.dot(k_fb.T)
    sigma_z = np.vstack((np.hstack((sigma_x, sigma_xu)), np.hstack((sigma_xu.T, sigma_u))))
    if taylor:
        sigma_zg = sigma_z.dot(jac_g.T)
        sigma_g = np.diag(var_g) + jac_g.dot(sigma_z).dot(jac_g.T)
    else:
        sigma_zg = np.zeros((n_s + n_u, n_s))
        sigma_g = np.diag(var_g)
    sigma_all = np.vstack((np.hstack((sigma_z, sigma_zg)), np.hstack((sigma_zg.T, sigma_g.T))))
    lin = np.hstack((a, b, np.eye(n_s)))
    mu_new = lin.dot(np.concatenate((mu_x, k_ff, mu_g)))
    return mu_new, lin.dot(sigma_all).dot(lin.T)


def multistep_moments_batch(model, mu_0, k_ff, k_fb, a, b, taylor=True):
    """uncertainty_propagation_casadi.py:88-146 / :149-207 with a leading T axis.
    mu_0 (T,n_s); k_ff (T,H,n_u); k_fb (T,H-1,n_u,n_s) -> mu_all (T,H,n_s), sigma_all (T,H,n_s,n_s)."""
    T, H, n_u = k_ff.shape
    n_s = mu_0.shape[1]
    mu_all = np.empty((T, H, n_s))
    sigma_all = np.empty((T, H, n_s, n_s))
    for t in range(T):
        mu, sig = mu_0[t], None
        for i in range(H):
            m, v, j = _predict_one(model, np.concatenate((mu, k_ff[t, i])))
            mu, sig = one_step_moments(mu, sig, k_ff[t, i], None if i == 0 else k_fb[t, i - 1], m, v, j, a, b, taylor)
            mu_all[t, i], sigma_all[t, i] = mu, sig
    return mu_all, sigma_all


# --------------------------------------------------------------------------- Monte-Carlo verification
def sample_from_gp(model, inp, eps):
    """ssm_gpy/gaussian_process.py:598-619 with GPy posterior_samples_f(full_cov=False): independent
    marginal draws per test input.  inp (n, D), eps (n, size, n_s) standard normal -> S (n, size, n_s)."""
    mu, var = gp_predict(inp, model["Z"], model["beta"], model["inv_K"], model["lengthscale"],
                         model["signal_var"], compute_gradients=False)
    return mu[:, None, :] + np.sqrt(var)[:, None, :] * eps


def mc_sample_n_step(model, x0, K, k, eps):
    """sampling_models.py:33-80.  x0 (n_s,1), K (n, n_u, n_s), k (n, n_u), eps (n, n_samples, n_s)
    -> S_all (n, n_samples, n_s)."""
    n, n_samples, n_s = eps.shape
    u0 = K[0].dot(x0) + k[0, :, None]
    inp0 = np.vstack((x0, u0)).T
    S_all = np.empty((n, n_samples, n_s))
    S = sample_from_gp(model, inp0, eps[0][None]).reshape(n_samples, n_s)
    S_all[0] = S
    for i in range(1, n):
        U = S.dot(K[i].T) + k[i][None, :]
        S = sample_from_gp(model, np.hstack((S, U)), eps[i][:, None, :]).reshape(n_samples, n_s)
        S_all[i] = S
    return S_all


def information_gain(Z, lengthscale, signal_var, noise_var_fixed):
    """ssm_gpy/gaussian_process.py:621-634: per output log det(I + K/sigma_n^2), K the noise-free training
    Gram matrix (GPy ``posterior._K``), sigma_n^2 the fixed Gaussian noise (incl. noise_diag, :252-253)."""
    out = []
    N = Z.shape[0]
    for d in range(len(signal_var)):
        Kd = rbf_kernel(Z, Z, signal_var[d], lengthscale[d])
        out.append(np.linalg.slogdet(np.eye(N) + Kd / noise_var_fixed[d])[1])
    return out


# --------------------------------------------------------------------------- data selection
def choose_datapoints_maxvar(x, y, m, init_idx, lengthscale, signal_var, noise_var):
    """Greedy part of ssm_gpy/gaussian_process.py:323-343 with fixed hyper-parameters: starting from the
    seed rows ``init_idx`` add, m - len(init_idx) times, the pool point with the largest summed posterior
    variance under the GP conditioned on the rows chosen so far (GPy ``set_XY(x_chosen, ...)``, :340-341).
    Refits from scratch every round.  Returns the chosen row indices in order."""
    chosen = [int(i) for i in init_idx]
    n_data = x.shape[0]
    while len(chosen) < m:
        beta, inv_K, _ = gp_fit(x[chosen], y[chosen], lengthscale, signal_var, noise_var)
        _, var = gp_predict(x, x[chosen], beta, inv_K, lengthscale, signal_var, compute_gradients=False)
        score = var.sum(axis=1)
        score[chosen] = -np.inf
        chosen.append(int(np.argmax(score)))
    return np.asarray(chosen)


# --------------------------------------------------------------------------- marginal likelihood (opt_hyp=True)
def gp_nll_grad(Z, y, kern_type, hyp, noise_var):
    """Negative log marginal likelihood of one output and its gradient with respect to every hyper-parameter
    (natural scale), the objective behind ``model_gp.optimize()`` in ssm_gpy/gaussian_process.py:249-250:
        nll = 1/2 y^T K_y^-1 y + 1/2 log det K_y + N/2 log 2pi ,  K_y = K + (noise_var + 1e-8) I
        d nll/d theta = 1/2 tr((K_y^-1 - alpha alpha^T) dK_y/d theta).
    dK/d theta is written out per kernel identifier (not through the packed family the device uses).
    Returns nll, {key: gradient array}."""
    N, D = Z.shape
    K = kernel_matrix(kern_type, hyp, Z, Z)
    Ky = K + (noise_var + GPY_JITTER) * np.eye(N)
    L = sla.cholesky(Ky, lower=True)
    alpha = sla.cho_solve((L, True), y)
    Kinv = sla.cho_solve((L, True), np.eye(N))
    nll = 0.5 * y.dot(alpha) + np.sum(np.log(np.diag(L))) + 0.5 * N * np.log(2 * np.pi)
    M = 0.5 * (Kinv - np.outer(alpha, alpha))
    grad = {"noise_variance": np.array([np.trace(M)])}

    def radial(st, x, ell):
        """kappa and d kappa / d ell_j for unit variance; x (N,d), ell (d,)"""
        diff2 = (x[:, None, :] - x[None, :, :]) ** 2                       # (N,N,d)
        r = np.sqrt(np.sum(diff2 / ell ** 2, axis=2))
        kap, g, _ = _stationary_radial(st, r)
        dl = -g[:, :, None] * diff2 / ell ** 3                               # d kappa/d ell_j = kappa'(r) dr/d ell_j
        return kap, dl

    if kern_type in ("rbf", "mat52"):
        ell = np.asarray(hyp["lengthscale"], np.float64).reshape(-1) * np.ones(D)
        kap, dl = radial(kern_type, Z, ell)
        grad["variance"] = np.array([np.sum(M * kap)])
        grad["lengthscale"] = float(hyp["variance"]) * np.einsum('ij,ijk->k', M, dl)
        return nll, grad
    st = "rbf" if kern_type == "lin_rbf" else "mat52"
    ell = np.asarray(hyp["prod.%s.lengthscale" % st], np.float64).reshape(-1)[:1]
    vs = float(hyp["prod.%s.variance" % st])
    vp = float(np.asarray(hyp["prod.linear.variances"]).reshape(-1)[0])
    kap, dl = radial(st, Z[:, 1:2], ell)
    zz = np.outer(Z[:, 1], Z[:, 1])
    grad["prod.%s.variance" % st] = np.array([np.sum(M * vp * zz * kap)])
    grad["prod.%s.lengthscale" % st] = np.array([np.sum(M * vp * vs * zz * dl[:, :, 0])])
    grad["prod.linear.variances"] = np.array([np.sum(M * vs * zz * kap)])
    grad["linear.variances"] = np.array([np.sum(M * np.outer(Z[:, j], Z[:, j])) for j in range(D)])
    return nll, grad
