#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What is executed from the reference (imported, never copied):
  * safe_exploration.utils_ellipsoid      (as-is)
  * safe_exploration.utils                (compute_remainder_overapproximations, sample_inside_polytope)
  * safe_exploration.gp_reachability      (onestep_reachability, multistep_reachability,
                                           lin_ellipsoid_safety_distance)
  * safe_exploration/ssm_gpy/gp_models_utils_casadi.py  (_k_rbf, _unscaled_dist, gp_pred), loaded by
    file path because the sub-package __init__ imports GPy.

casadi is not installed.  utils.py needs the *name* ``casadi.reshape`` at import; the
gp_models_utils_casadi formulas call six casadi array functions (mtimes, exp, sum2, repmat, sqrt,
SX(n)).  A throw-away numpy-backed module named ``casadi`` is written to a temp dir for THIS script
only so that those reference formulas can be evaluated on numbers.  GPy itself is absent, so the GPy
boundary stays unpinned (see oracle/oracle_np.py header).

The GP posterior state (beta, inv_K) fed to the reference formulas comes from oracle.gp_fit; the
(mu, var, jac) fed to the reference reachability functions come from oracle.gp_predict and are stored
in the fixtures so the ellipsoid kernel can be tested in isolation from the GP kernels.
"""
import importlib.util
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

CASADI_SHIM = '''
import numpy as _np
import numpy as np          # the reference relies on `from casadi import *` exporting np
def reshape(a, s): return _np.reshape(a, s, order="F")
def mtimes(a, b): return _np.dot(a, b)
def exp(a): return _np.exp(a)
def sqrt(a): return _np.sqrt(a)
def sum2(a): return _np.sum(a, axis=1, keepdims=True)
def repmat(a, n, m=1): return _np.tile(_np.atleast_2d(a), (n, m))
def vertcat(*a): return _np.vstack([x for x in a if _np.size(x)])
def horzcat(*a): return _np.hstack([x for x in a if _np.size(x)])
class SX(object):
    def __new__(cls, *a):
        return _np.zeros((1, 1)) if len(a) < 2 else _np.zeros(a)
def diag(a):
    a = _np.asarray(a)
    return _np.diagflat(a) if (a.ndim == 1 or 1 in a.shape) else _np.diag(a)[:, None]
class MX(object):
    eye = staticmethod(_np.eye)
    zeros = staticmethod(lambda *a: _np.zeros(a))
class Function(object):
    def __init__(self, *a, **k): raise NotImplementedError("numeric shim only")
'''


def _import_reference():
    shim_dir = tempfile.mkdtemp(prefix="casadi_shim_")
    with open(os.path.join(shim_dir, "casadi.py"), "w") as f:
        f.write(CASADI_SHIM)
    sys.path.insert(0, shim_dir)
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    warnings.simplefilter("ignore")
    from safe_exploration import utils_ellipsoid, utils, gp_reachability
    global ref_prop
    from safe_exploration import uncertainty_propagation_casadi as ref_prop
    spec = importlib.util.spec_from_file_location(
        "ref_gp_models_utils_casadi",
        os.path.join(REF, "safe_exploration/ssm_gpy/gp_models_utils_casadi.py"))
    gpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gpu)
    return utils_ellipsoid, utils, gp_reachability, gpu


def _save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, {k: np.shape(v) for k, v in arrs.items()})


def main():
    ue, ut, gr, gpu = _import_reference()
    from oracle import oracle_np as orc

    # ------------------------------------------------------------------ 1. utils_ellipsoid
    rng = np.random.default_rng(11)
    out = {}
    for n in (2, 3, 4, 8):
        ub = rng.uniform(0.05, 0.5, n)
        out["rect_ub_%d" % n] = ub
        out["rect_q_%d" % n] = ue.ellipsoid_from_rectangle(ub)
        A1, A2 = rng.standard_normal((n, n)), rng.standard_normal((n, n))
        q1, q2 = A1.dot(A1.T) + 0.1 * np.eye(n), A2.dot(A2.T) + 0.2 * np.eye(n)
        p1, p2 = rng.standard_normal((n, 1)), rng.standard_normal((n, 1))
        ps, qs = ue.sum_two_ellipsoids(p1, q1, p2, q2)
        pc, qc = ue.sum_two_ellipsoids(p1, q1, p2, q2, c=0.7)
        out.update({"sum_p1_%d" % n: p1, "sum_q1_%d" % n: q1, "sum_p2_%d" % n: p2,
                    "sum_q2_%d" % n: q2, "sum_p_%d" % n: ps, "sum_q_%d" % n: qs,
                    "sumc_q_%d" % n: qc})
        # sum_ellipsoids with direction l, 3 and 4 ellipsoids
        for k in (3, 4):
            pp = rng.standard_normal((k, n))
            qq = np.empty((k, n, n))
            for i in range(k):
                B = rng.standard_normal((n, n))
                qq[i] = B.dot(B.T) + 0.1 * np.eye(n)
            l = rng.standard_normal((n, 1))
            pn, qn = ue.sum_ellipsoids(pp, qq, l)
            out.update({"msum_p_in_%d_%d" % (k, n): pp, "msum_q_in_%d_%d" % (k, n): qq,
                        "msum_l_%d_%d" % (k, n): l, "msum_p_%d_%d" % (k, n): pn,
                        "msum_q_%d_%d" % (k, n): qn})
        S = rng.standard_normal((7, n))
        out["dist_s_%d" % n] = S
        out["dist_d_%d" % n] = ue.distance_to_center(S, p1, q1)
        out["inside_%d" % n] = ue.sample_inside_ellipsoid(S, p1, q1, c=3.0)
    # known answers quoted by the reference tests / __main__
    out["known_rect"] = ue.ellipsoid_from_rectangle([0.1, 0.2, 0.3])          # diag(.03,.12,.27)
    out["known_dist"] = ue.distance_to_center(np.array([[1 + np.sqrt(2), 1.0]]),
                                              np.array([[1.0], [1.0]]), 4 * np.eye(2) / 2)
    _save("ellipsoid.npz", **out)

    # ------------------------------------------------------------------ 2. remainder over-approximation
    out = {}
    for tag, (n_s, n_u) in zip("1234", [(2, 1), (3, 2), (5, 4), (8, 3)]):
        np.random.seed(0)                     # the reference test's own recipe (test_utils_casadi.py:99-120)
        x_0 = np.random.rand(n_s, n_s)
        q = np.dot(x_0, x_0.T) + 0.1 * np.eye(n_s)
        k_fb = np.random.randn(n_u, n_s)
        l_mu = np.array([.1] * n_s)
        l_sigma = np.array([.1] * n_s)
        u_mu, u_sigma = ut.compute_remainder_overapproximations(q, k_fb, l_mu, l_sigma)
        assert np.all(np.imag(u_mu) == 0) and np.all(np.imag(u_sigma) == 0)
        out.update({"q_" + tag: q, "k_fb_" + tag: k_fb, "l_mu_" + tag: l_mu, "l_sigma_" + tag: l_sigma,
                    "u_mu_" + tag: np.real(u_mu), "u_sigma_" + tag: np.real(u_sigma)})
        u0m, u0s = ut.compute_remainder_overapproximations(q, np.zeros((n_u, n_s)), l_mu, l_sigma)
        out["u_mu_k0_" + tag] = np.real(u0m)
        out["u_sigma_k0_" + tag] = np.real(u0s)
    # polytope known answer (test_utils.py:12-29)
    x = np.array([[0.1, 0.15], [0.0, 0.0], [.5, .15]])
    a = np.vstack((np.eye(2), -np.eye(2), -np.eye(2)))
    b = np.array([.4, .2, .3, .2, .3, .2])[:, None]
    out.update({"poly_x": x, "poly_a": a, "poly_b": b, "poly_res": ut.sample_inside_polytope(x, a, b)})
    _save("remainder.npz", **out)

    # ------------------------------------------------------------------ 3. GP formulas vs reference gp_pred
    def gp_case(name, seed, N, n_s, n_u, T):
        syn = orc.make_synthetic(seed, N, n_s, n_u, T)
        beta, inv_K, chol = orc.gp_fit(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"],
                                       syn["noise_var"])
        x_new = np.hstack((syn["p"], syn["k_ff"]))
        mu, var, jac = orc.gp_predict(x_new, syn["Z"], beta, inv_K, syn["lengthscale"],
                                      syn["signal_var"], True)
        ref_mu = np.empty_like(mu)
        ref_var = np.empty_like(var)
        ref_K = []
        for d in range(n_s):
            hyp = {"lengthscale": syn["lengthscale"][d], "variance": syn["signal_var"][d]}
            kern = gpu._get_kernel_function("rbf", hyp)
            ref_K.append(np.asarray(kern(x_new, y=syn["Z"])))
            for t in range(T):       # gp_pred is used single-query by the reference (N==1 assert, :164)
                m_t, s_t = gpu.gp_pred(x_new[t:t + 1], kern, beta[:, d:d + 1], syn["Z"], inv_K[d], True)
                ref_mu[t, d] = np.asarray(m_t).item()
                ref_var[t, d] = np.asarray(s_t).item()
        # oracle == reference's formulas on numbers (before the GPy-style 1e-15 clip, which never binds here)
        assert np.allclose(ref_mu, mu, rtol=1e-12, atol=1e-13), np.abs(ref_mu - mu).max()
        assert np.allclose(ref_var, var, rtol=0, atol=1e-12), np.abs(ref_var - var).max()
        for d in range(n_s):
            assert np.allclose(ref_K[d], orc.rbf_kernel(x_new, syn["Z"], syn["signal_var"][d],
                                                        syn["lengthscale"][d]), rtol=1e-13, atol=0)
        jv, hm = orc.gp_linearize_extras(x_new[0], syn["Z"], beta, inv_K, syn["lengthscale"],
                                         syn["signal_var"])
        _save(name, Z=syn["Z"], Y=syn["Y"], lengthscale=syn["lengthscale"],
              signal_var=syn["signal_var"], noise_var=syn["noise_var"], x_new=x_new, beta=beta,
              mu=mu, var=var, jac=jac, ref_mu=ref_mu, ref_var=ref_var, ref_kstar0=ref_K[0],
              jac_var0=jv, hess_mu0=hm)
        return syn, beta, inv_K

    gp_case("gp_pend.npz", 101, 60, 2, 1, 33)
    gp_case("gp_cart.npz", 102, 90, 4, 1, 20)

    # ------------------------------------------------------------------ 3b. non-RBF kernel types (8(f).1)
    for kt in ("mat52", "lin_rbf", "lin_mat52"):
        rng = np.random.default_rng({"mat52": 301, "lin_rbf": 302, "lin_mat52": 303}[kt])
        n_s, n_u, N, T = 2, 1, 50, 21
        D = n_s + n_u
        Z = rng.uniform(-1, 1, (N, D))
        Y = np.sin(2.0 * Z.dot(rng.standard_normal((D, n_s)))) + 0.05 * rng.standard_normal((N, n_s))
        x_new = 0.5 * rng.standard_normal((T, D))
        hyp = [orc.make_hyp(kt, rng, D) for _ in range(n_s)]
        noise = np.full(n_s, 1e-2 + 1e-5)
        beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * n_s, hyp, noise)
        mu, var = orc.gp_predict_k(x_new, Z, beta, inv_K, [kt] * n_s, hyp)
        ref_mu = np.empty_like(mu); ref_var = np.empty_like(var)
        ref_K0 = None
        for d in range(n_s):
            kern = gpu._get_kernel_function(kt, hyp[d])
            Kref = np.asarray(kern(x_new, y=Z))
            assert np.allclose(Kref, orc.kernel_matrix(kt, hyp[d], x_new, Z), rtol=1e-12, atol=1e-14), kt
            if d == 0:
                ref_K0 = Kref      # (k(Z,Z) is not taken from the reference: its unclipped sqrt(r2) gives NaN
                #                    on the diagonal whenever -2xy+x^2+y^2 rounds below zero)
            for t in range(T):
                m_t, s_t = gpu.gp_pred(x_new[t:t + 1], kern, beta[:, d:d + 1], Z, inv_K[d], True)
                ref_mu[t, d] = np.asarray(m_t).item()
                ref_var[t, d] = np.asarray(s_t).item()
        assert np.allclose(ref_mu, mu, rtol=1e-11, atol=1e-12), kt
        assert np.allclose(ref_var, var, rtol=0, atol=1e-10), (kt, np.abs(ref_var - var).max())
        flat = {}
        for d in range(n_s):
            for k, v in hyp[d].items():
                flat["hyp%d_%s" % (d, k)] = np.asarray(v)
        _save("kern_%s.npz" % kt, Z=Z, Y=Y, x_new=x_new, noise_var=noise, beta=beta, mu=mu, var=var,
              ref_mu=ref_mu, ref_var=ref_var, ref_kstar0=ref_K0,
              jac_fd=orc.gp_mean_jacobian_fd(x_new, Z, beta, [kt] * n_s, hyp), **flat)

    # ------------------------------------------------------------------ 4. reachability through the reference
    def reach_case(name, seed, N, n_s, n_u, T, H, l_mu, l_sigma, c_safety, a_scale=1.0, sf2=1.0):
        syn = orc.make_synthetic(seed, N, n_s, n_u, T, sf2=sf2)
        beta, inv_K, _ = orc.gp_fit(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"],
                                    syn["noise_var"])
        model = dict(Z=syn["Z"], beta=beta, inv_K=inv_K, lengthscale=syn["lengthscale"],
                     signal_var=syn["signal_var"])
        rng = np.random.default_rng(seed + 1000)
        a_lin = a_scale * np.eye(n_s) + 0.05 * rng.standard_normal((n_s, n_s))
        b_lin = 0.1 * rng.standard_normal((n_s, n_u))

        # GP outputs per query, computed ONCE and replayed to the reference bit-for-bit (a batched and a
        # single-row BLAS call differ in the last bits of the cancelling variance)
        x = np.hstack((syn["p"], syn["k_ff"]))
        mu = np.empty((T, n_s)); var = np.empty((T, n_s)); jac = np.empty((T, n_s, n_s + n_u))
        cache = {}
        for t in range(T):
            mu[t], var[t], jac[t] = orc._predict_one(model, x[t])
            cache[x[t].tobytes()] = t

        def ssm(states, actions):              # 3-tuple contract of SimpleGPModel.__call__ (A5)
            z = np.ascontiguousarray(np.hstack((np.asarray(states), np.asarray(actions)))[0])
            t = cache.get(z.tobytes())
            if t is not None:
                return mu[t][:, None].copy(), var[t][:, None].copy(), jac[t].copy()
            m, v, j = orc._predict_one(model, z)
            return m[:, None], v[:, None], j

        res = {k: syn[k] for k in ("Z", "Y", "lengthscale", "signal_var", "noise_var", "p", "k_ff",
                                   "k_fb", "Q")}
        res.update(mu=mu, var=var, jac=jac, l_mu=l_mu, l_sigma=l_sigma, c_safety=c_safety,
                   a_lin=a_lin, b_lin=b_lin)
        for tag, (aa, bb) in {"id": (None, None), "lin": (a_lin, b_lin)}.items():
            p1_pt = np.empty((T, n_s)); q1_pt = np.empty((T, n_s, n_s))
            p1_el = np.empty((T, n_s)); q1_el = np.empty((T, n_s, n_s))
            for t in range(T):
                pp, qq = gr.onestep_reachability(syn["p"][t][:, None], ssm, syn["k_ff"][t][:, None],
                                                 l_mu, l_sigma, None, None, c_safety, 0, aa, bb)
                p1_pt[t], q1_pt[t] = pp[:, 0], qq
                pp, qq = gr.onestep_reachability(syn["p"][t][:, None], ssm, syn["k_ff"][t][:, None],
                                                 l_mu, l_sigma, syn["Q"][t], syn["k_fb"][t], c_safety,
                                                 0, aa, bb)
                assert np.all(np.imag(qq) == 0)
                p1_el[t], q1_el[t] = pp[:, 0], np.real(qq)
            res.update({"p1_point_" + tag: p1_pt, "q1_point_" + tag: q1_pt,
                        "p1_ell_" + tag: p1_el, "q1_ell_" + tag: q1_el})
        # multistep chains (workload generator of uncertainty_propagation_runner.py:32-33)
        Tm = min(T, 6)
        k_fb_m = 0.1 * rng.standard_normal((Tm, H - 1, n_u, n_s))
        k_ff_m = 0.1 * rng.standard_normal((Tm, H, n_u))
        p0_m = 0.1 * rng.standard_normal((Tm, n_s))
        p_all = np.empty((Tm, H, n_s)); q_all = np.empty((Tm, H, n_s, n_s))
        p_all_q0 = np.empty((Tm, H, n_s)); q_all_q0 = np.empty((Tm, H, n_s, n_s))
        for t in range(Tm):
            _, _, pa, qa = gr.multistep_reachability(p0_m[t][:, None], ssm, k_fb_m[t], k_ff_m[t], l_mu,
                                                     l_sigma, None, c_safety, 0, a_lin, b_lin, None)
            p_all[t], q_all[t] = pa, qa
            _, _, pa, qa = gr.multistep_reachability(p0_m[t][:, None], ssm, k_fb_m[t], k_ff_m[t], l_mu,
                                                     l_sigma, syn["Q"][t], c_safety, 0, a_lin, b_lin,
                                                     syn["k_fb"][t])
            p_all_q0[t], q_all_q0[t] = pa, qa
        assert np.all(np.isfinite(q_all)) and np.all(np.isfinite(q_all_q0)), "chain diverged"
        print(name, "max |q| along chain", np.abs(q_all).max(), np.abs(q_all_q0).max())
        res.update(ms_k_fb=k_fb_m, ms_k_ff=k_ff_m, ms_p0=p0_m, ms_p_all=p_all, ms_q_all=q_all,
                   ms_p_all_q0=p_all_q0, ms_q_all_q0=q_all_q0)
        # safety distance on the ellipsoid-branch results
        h_mat = np.vstack((np.eye(n_s), -np.eye(n_s)))
        h_vec = np.ones((2 * n_s, 1))
        d = np.empty((T, 2 * n_s))
        for t in range(T):
            d[t] = gr.lin_ellipsoid_safety_distance(res["p1_ell_id"][t][:, None], res["q1_ell_id"][t],
                                                    h_mat, h_vec, c_safety)[:, 0]
        res.update(h_mat=h_mat, h_vec=h_vec, d_safety=d)
        _save(name, **res)

    reach_case("reach_pend.npz", 201, 60, 2, 1, 24, 15, np.array([0.05, 0.02]), np.array([0.05, 0.02]), 2.0, a_scale=0.8, sf2=0.01)
    reach_case("reach_cart.npz", 202, 90, 4, 1, 16, 15, np.array([0.05] * 4), np.array([0.05] * 4), 2.0, a_scale=0.5, sf2=0.01)
    reach_case("reach_n3u2.npz", 203, 40, 3, 2, 8, 3, np.array([0.01] * 3), np.array([0.02] * 3), 1.5)

    # ------------------------------------------------------------------ 4a'. GP input transform (t_z_gp / a_gp_inp_x)
    # gp_reachability_casadi.onestep_reachability(..., t_z_gp) (:60-61,85,94-97) evaluates the GP at t_z_gp @ state and
    # chain-rules its state Jacobian; with the SSM wrapped accordingly the reference's NUMERIC onestep/multistep functions
    # compute exactly that (H = a + jac[:, :n_in] T + (jac[:, n_in:] + b) K).  The journal cart-pole configuration drops
    # the cart position (defaultconfig_episode.py:28,44): n_s = 4, n_x_in = 3, D = 4.  The moment propagation takes
    # a_gp_inp_x directly (uncertainty_propagation_casadi.py:40-47,60).
    def tz_case(name, seed, N, n_s, n_xin, n_u, T, H, tz, l_mu, l_sigma, c_safety, a_scale):
        rng = np.random.default_rng(seed)
        Dg = n_xin + n_u
        Z = rng.uniform(-1, 1, (N, Dg))
        Wd = rng.standard_normal((n_s, Dg))
        sf2 = 0.01
        Y = np.sqrt(sf2) * (np.sin(2.0 * Z.dot(Wd.T)) + 0.05 * rng.standard_normal((N, n_s)))
        ls = rng.uniform(0.5, 1.5, (n_s, Dg))
        signal_var = np.full(n_s, sf2)
        noise_var = np.full(n_s, 1e-2 * sf2 + 1e-5)
        beta, inv_K, _ = orc.gp_fit(Z, Y, ls, signal_var, noise_var)
        model = dict(Z=Z, beta=beta, inv_K=inv_K, lengthscale=ls, signal_var=signal_var)
        a_lin = a_scale * np.eye(n_s) + 0.05 * rng.standard_normal((n_s, n_s))
        b_lin = 0.1 * rng.standard_normal((n_s, n_u))
        p = 0.3 * rng.standard_normal((T, n_s))
        k_ff = 0.1 * rng.standard_normal((T, n_u))
        k_fb = 0.1 * rng.standard_normal((T, n_u, n_s))
        A = rng.standard_normal((T, n_s, n_s))
        Q = 0.01 * np.einsum('tij,tkj->tik', A, A) + 0.01 * np.eye(n_s)[None]

        def ssm_full(states, actions):        # full state in, GP evaluated at tz @ state, Jacobian w.r.t. [state; action]
            x = tz.dot(np.asarray(states)[0])
            m, v, j = orc._predict_one(model, np.hstack((x, np.asarray(actions)[0])))
            return m[:, None], v[:, None], np.hstack((j[:, :n_xin].dot(tz), j[:, n_xin:]))

        def ssm_gp(states, actions):          # what the casadi-side functions call: the GP on its own inputs
            m, v, j = orc._predict_one(model, np.hstack((np.asarray(states)[0], np.asarray(actions)[0])))
            return m[:, None], v[:, None], j

        res = dict(Z=Z, Y=Y, lengthscale=ls, signal_var=signal_var, noise_var=noise_var, tz=tz, p=p, k_ff=k_ff, k_fb=k_fb,
                   Q=Q, a_lin=a_lin, b_lin=b_lin, l_mu=l_mu, l_sigma=l_sigma, c_safety=c_safety)
        p1_pt = np.empty((T, n_s)); q1_pt = np.empty((T, n_s, n_s)); p1_el = np.empty((T, n_s)); q1_el = np.empty((T, n_s, n_s))
        for t in range(T):
            pp, qq = gr.onestep_reachability(p[t][:, None], ssm_full, k_ff[t][:, None], l_mu, l_sigma, None, None,
                                             c_safety, 0, a_lin, b_lin)
            p1_pt[t], q1_pt[t] = pp[:, 0], qq
            pp, qq = gr.onestep_reachability(p[t][:, None], ssm_full, k_ff[t][:, None], l_mu, l_sigma, Q[t], k_fb[t],
                                             c_safety, 0, a_lin, b_lin)
            assert np.all(np.imag(qq) == 0)
            p1_el[t], q1_el[t] = pp[:, 0], np.real(qq)
        res.update(p1_point=p1_pt, q1_point=q1_pt, p1_ell=p1_el, q1_ell=q1_el)
        Tm = min(T, 5)
        k_fb_m = 0.1 * rng.standard_normal((Tm, H - 1, n_u, n_s))
        k_ff_m = 0.1 * rng.standard_normal((Tm, H, n_u))
        p0_m = 0.1 * rng.standard_normal((Tm, n_s))
        p_all = np.empty((Tm, H, n_s)); q_all = np.empty((Tm, H, n_s, n_s))
        for t in range(Tm):
            _, _, pa, qa = gr.multistep_reachability(p0_m[t][:, None], ssm_full, k_fb_m[t], k_ff_m[t], l_mu, l_sigma, None,
                                                     c_safety, 0, a_lin, b_lin, None)
            p_all[t], q_all[t] = pa, qa
        assert np.all(np.isfinite(q_all)), "chain diverged"
        res.update(ms_k_fb=k_fb_m, ms_k_ff=k_ff_m, ms_p0=p0_m, ms_p_all=p_all, ms_q_all=q_all)
        # moment propagation through the reference's own builders with a_gp_inp_x = tz
        for tag, fn in (("taylor", ref_prop.multi_step_taylor_symbolic), ("meaneq", ref_prop.mean_equivalent_multistep)):
            mu_all = np.empty((Tm, H, n_s)); sig_all = np.empty((Tm, H, n_s, n_s))
            for t in range(Tm):
                m_, s_, _ = fn(p0_m[t][:, None], ssm_gp, k_ff_m[t], list(k_fb_m[t]), None, a_lin, b_lin, tz)
                mu_all[t], sig_all[t] = np.asarray(m_), np.asarray(s_).reshape(H, n_s, n_s)
            res["mu_" + tag], res["sigma_" + tag] = mu_all, sig_all
        _save(name, **res)

    tz_case("reach_tz_cart.npz", 271, 70, 4, 3, 1, 10, 8, np.hstack((np.zeros((3, 1)), np.eye(3))),
            np.array([0.05] * 4), np.array([0.05] * 4), 2.0, 0.5)
    tz_case("reach_tz_n3.npz", 272, 50, 3, 2, 2, 6, 4, np.random.default_rng(5).standard_normal((2, 3)) * 0.7,
            np.array([0.01] * 3), np.array([0.02] * 3), 1.5, 0.6)

    # ------------------------------------------------------------------ 4b. the reference tests' canonical scenarios
    # on the reference's OWN data files (copied as data: tests/golden/ref_invpend_data.npz = test/invpend_data.npz,
    # ref_data_cartpole.npz = test/data_cartpole.npz).  test_gp_reachability_casadi.py:30-67: seed 125, m = 50 random
    # points, c_safety 2, L = 0.001, q = .2 [[.5,.2],[.2,.65]], random a, b, k_fb, k_ff, p = .1 randn, T = 3 chain.
    # test_safempc.py:56-69,131-133: cart-pole data, 20 points, q = 0.1 I, the file's own linear model a, b.
    # The hyper-parameters are FIXED here (the reference optimises them inside GPy, which cannot run).
    def ref_scenario(name, data_file, n_s, n_u, m, seed, Lc, q0, hyp_ls, hyp_sf2, hyp_sn2, use_file_ab):
        data = np.load(os.path.join(HERE, data_file))
        X, y = data["X"], data["y"]
        np.random.seed(seed)
        a_lin = np.random.rand(n_s, n_s) if not use_file_ab else np.asarray(data["a"], dtype=np.float64)
        b_lin = np.random.rand(n_s, n_u) if not use_file_ab else np.asarray(data["b"], dtype=np.float64)
        idx = np.random.choice(X.shape[0], size=m, replace=False)
        Z, Yz = X[idx], y[idx]
        k_fb = np.random.rand(n_u, n_s)
        k_ff = np.random.rand(n_u, 1)
        p = .1 * np.random.randn(n_s, 1)
        T = 3
        u_0 = .2 * np.random.randn(n_u, 1)
        k_fb_0 = np.random.randn(T - 1, n_s * n_u)
        k_ff_m = np.random.randn(T - 1, n_u)
        k_ff_all = np.vstack((u_0.T, k_ff_m))
        k_fb_apply = k_fb_0.reshape(-1, n_u, n_s) + k_fb[None]
        D = n_s + n_u
        ls = np.full((n_s, D), hyp_ls); sf2 = np.full(n_s, hyp_sf2); noise = np.full(n_s, hyp_sn2 + 1e-5)
        beta, inv_K, _ = orc.gp_fit(Z, Yz, ls, sf2, noise)
        model = dict(Z=Z, beta=beta, inv_K=inv_K, lengthscale=ls, signal_var=sf2)

        def ssm(states, actions):
            z = np.ascontiguousarray(np.hstack((np.asarray(states), np.asarray(actions)))[0])
            mm, vv, jj = orc._predict_one(model, z)
            return mm[:, None], vv[:, None], jj

        L = np.array([Lc] * n_s)
        out = dict(Z=Z, Y=Yz, lengthscale=ls, signal_var=sf2, noise_var=noise, a_lin=a_lin, b_lin=b_lin, k_fb=k_fb,
                   k_ff=k_ff, p=p, q0=q0, L=L, k_ff_all=k_ff_all, k_fb_apply=k_fb_apply)
        for tag, (aa, bb) in {"id": (None, None), "lin": (a_lin, b_lin)}.items():
            pp, qq = gr.onestep_reachability(p, ssm, k_ff, L, L, q0, k_fb, 2, 0, a=aa, b=bb)
            out["p1_ell_" + tag], out["q1_ell_" + tag] = pp, np.real(qq)
            pp, qq = gr.onestep_reachability(p, ssm, k_ff, L, L, None, k_fb, 2, 0, a=aa, b=bb)
            out["p1_pt_" + tag], out["q1_pt_" + tag] = pp, np.real(qq)
            _, _, pa, qa = gr.multistep_reachability(p, ssm, k_fb_apply, k_ff_all, L, L, None, 2, 0, aa, bb, None)
            assert np.all(np.isfinite(qa))
            out["ms_p_" + tag], out["ms_q_" + tag] = pa, qa
        _save(name, **out)

    ref_scenario("scen_invpend.npz", "ref_invpend_data.npz", 2, 1, 50, 125, 0.001,
                 .2 * np.array([[.5, .2], [.2, .65]]), 0.6, 0.05, 1e-3, False)
    ref_scenario("scen_cartpole.npz", "ref_data_cartpole.npz", 4, 1, 20, 125, 0.001, 0.1 * np.eye(4), 1.0, 0.05,
                 1e-3, True)

    # ------------------------------------------------------------------ 4c. Gaussian moment propagation (8(f).4)
    # the reference's symbolic graph builders evaluated on numbers (casadi array functions -> numpy)
    def moments_case(name, seed, N, n_s, n_u, T, H):
        syn = orc.make_synthetic(seed, N, n_s, n_u, T, sf2=0.01)
        beta, inv_K, _ = orc.gp_fit(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        model = dict(Z=syn["Z"], beta=beta, inv_K=inv_K, lengthscale=syn["lengthscale"], signal_var=syn["signal_var"])
        rng = np.random.default_rng(seed + 5)
        a_lin = 0.8 * np.eye(n_s) + 0.05 * rng.standard_normal((n_s, n_s))
        b_lin = 0.1 * rng.standard_normal((n_s, n_u))
        k_ff = 0.1 * rng.standard_normal((T, H, n_u))
        k_fb = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
        mu0 = 0.1 * rng.standard_normal((T, n_s))

        def ssm(states, actions):
            z = np.ascontiguousarray(np.hstack((np.asarray(states), np.asarray(actions)))[0])
            m, v, j = orc._predict_one(model, z)
            return m[:, None], v[:, None], j

        res = {k: syn[k] for k in ("Z", "Y", "lengthscale", "signal_var", "noise_var")}
        res.update(a_lin=a_lin, b_lin=b_lin, k_ff=k_ff, k_fb=k_fb, mu0=mu0)
        for tag, fn in (("taylor", ref_prop.multi_step_taylor_symbolic), ("meaneq", ref_prop.mean_equivalent_multistep)):
            mu_all = np.empty((T, H, n_s)); sig_all = np.empty((T, H, n_s, n_s)); gv_all = np.empty((T, H, n_s))
            for t in range(T):
                m_, s_, g_ = fn(mu0[t][:, None], ssm, k_ff[t], list(k_fb[t]), None, a_lin, b_lin)
                mu_all[t], sig_all[t] = np.asarray(m_), np.asarray(s_).reshape(H, n_s, n_s)
                if tag == "meaneq":      # (the Taylor variant returns sigma_g as a MATRIX after step 0, :69-87)
                    gv_all[t] = np.asarray(g_)
            res["mu_" + tag], res["sigma_" + tag] = mu_all, sig_all
            if tag == "meaneq":
                res["gpvar_meaneq"] = gv_all
            om, osig = orc.multistep_moments_batch(model, mu0, k_ff, k_fb, a_lin, b_lin, tag == "taylor")
            assert np.allclose(om, mu_all, rtol=1e-12, atol=1e-14) and np.allclose(osig, sig_all, rtol=1e-11, atol=1e-16)
        _save(name, **res)

    moments_case("moments_pend.npz", 401, 60, 2, 1, 5, 6)
    moments_case("moments_cart.npz", 402, 80, 4, 1, 4, 5)

    # ------------------------------------------------------------------ 6. Monte-Carlo verification
    # The reference's MonteCarloSafetyVerification driven with a stand-in GP whose sample_from_gp replays
    # stored standard-normal draws through the oracle posterior: pins the closed-loop bookkeeping
    # (K[0] on x0, K[i] on the particles of step i-1, shapes) and inside_ellipsoid_ratio.
    from safe_exploration import sampling_models as ref_mc

    def mc_case(name, seed, N, n_s, n_u, n, n_samples):
        syn = orc.make_synthetic(seed, N, n_s, n_u, 4, sf2=0.01)
        beta, inv_K, _ = orc.gp_fit(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        model = dict(Z=syn["Z"], beta=beta, inv_K=inv_K, lengthscale=syn["lengthscale"], signal_var=syn["signal_var"])
        rng = np.random.default_rng(seed + 9)
        eps = rng.standard_normal((n, n_samples, n_s))
        K = 0.1 * rng.standard_normal((n, n_u, n_s))
        k = 0.1 * rng.standard_normal((n, n_u))
        x0 = 0.1 * rng.standard_normal((n_s, 1))

        class ReplayGP(object):
            def __init__(self):
                self.n_s, self.n_u, self.calls = n_s, n_u, 0

            def sample_from_gp(self, inp, size=10):
                e = eps[self.calls][None] if self.calls == 0 else eps[self.calls][:, None, :]
                self.calls += 1
                assert e.shape == (inp.shape[0], size, n_s)
                return orc.sample_from_gp(model, np.asarray(inp, dtype=np.float64), e)

        mc = ref_mc.MonteCarloSafetyVerification(ReplayGP())
        S, S_all = mc.sample_n_step(x0, K, k, n, n_samples)
        assert np.allclose(orc.mc_sample_n_step(model, x0, K, k, eps), S_all, rtol=1e-13, atol=1e-15)
        # ellipsoids around the particle clouds, sized so that a fraction of the particles falls outside
        p = S_all.mean(axis=1)
        Q = np.empty((n, n_s, n_s)); ratio = np.empty(n); inside = np.empty((n, n_samples), dtype=bool)
        for i in range(n):
            c = np.cov(S_all[i].T) + 1e-12 * np.eye(n_s)
            Q[i] = (1.0 + 0.5 * i) * c * 2.0
            r, rb = mc.inside_ellipsoid_ratio(S_all[i][None], Q[i][None], p[i][None])   # (the reference's float() needs n == 1)
            ratio[i], inside[i] = r, rb[0].astype(bool)
        res = {kk: syn[kk] for kk in ("Z", "Y", "lengthscale", "signal_var", "noise_var")}
        res.update(eps=eps, K=K, k=k, x0=x0, S_all=S_all, S_last=S, ell_p=p, ell_q=Q, ratio=ratio, inside=inside)
        _save(name, **res)

    mc_case("mc_pend.npz", 501, 60, 2, 1, 4, 64)      # n_u = 1: the reference's repmat of k[i] is only shaped right there
    mc_case("mc_cart.npz", 502, 80, 4, 1, 3, 48)

    # ------------------------------------------------------------------ 7. true-system roll-outs vs ellipsoids
    # gp_reachability.simulate_trajectory / verify_trajectory_safety / trajectory_inside_ellipsoid driven with a
    # stand-in environment (the reference's environments need scipy.integrate setups that are out of scope).
    np.bool = bool                                # gp_reachability.py:350 still uses the removed alias

    class ToyEnv(object):
        n_s, n_u = 2, 1

        def __init__(self, A, Bm):
            self.A, self.Bm = A, Bm

        def simulate_onestep(self, x, u):
            x = np.asarray(x, dtype=np.float64).reshape(-1)
            u = np.asarray(u, dtype=np.float64).reshape(-1)
            xn = self.A.dot(x) + self.Bm.dot(u) + 0.05 * np.sin(x)
            return xn, None

    rng = np.random.default_rng(77)
    A = np.array([[0.9, 0.1], [-0.05, 0.85]]); Bm = np.array([[0.0], [0.3]])
    env = ToyEnv(A, Bm)
    n = 6
    p_0 = 0.2 * rng.standard_normal(2)
    k_ff = 0.2 * rng.standard_normal((n, 1))
    k_fb = 0.3 * rng.standard_normal((n - 1, 2))
    p_all = 0.15 * rng.standard_normal((n, 2))
    q_all = np.empty((n, 4))
    for i in range(n):
        Bq = rng.standard_normal((2, 2))
        q_all[i] = ((0.02 + 0.03 * i) * (Bq.dot(Bq.T) + np.eye(2))).reshape(-1)
    x_all = gr.simulate_trajectory(env, p_0, k_fb, k_ff, p_all[:n - 1])
    inside = gr.trajectory_inside_ellipsoid(env, p_0, p_all, q_all, k_fb, k_ff)
    h_mat = np.vstack((np.eye(2), -np.eye(2)))
    ok_wide, _ = gr.verify_trajectory_safety(env, p_0, k_fb, k_ff, p_all[:n - 1], h_mat, np.full((4, 1), 5.0), h_mat,
                                             np.full((4, 1), 5.0))
    lim = float(np.abs(x_all[-1]).max()) * 0.9
    ok_tight, _ = gr.verify_trajectory_safety(env, p_0, k_fb, k_ff, p_all[:n - 1], h_mat, np.full((4, 1), lim))
    _save("traj.npz", A=A, Bm=Bm, p_0=p_0, k_ff=k_ff, k_fb=k_fb, p_all=p_all, q_all=q_all, x_all=x_all,
          inside=np.asarray(inside, dtype=bool), h_mat=h_mat, ok_wide=bool(ok_wide), ok_tight=bool(ok_tight), lim=lim)

    # ------------------------------------------------------------------ 8. LQR gains (utils.dlqr, :20-35)
    rng = np.random.default_rng(3)
    out = {}
    for tag, (n, m) in zip("ab", [(2, 1), (4, 2)]):
        a_ = np.eye(n) + 0.1 * rng.standard_normal((n, n)); b_ = rng.standard_normal((n, m))
        q_ = np.eye(n) * 2.0; r_ = np.eye(m) * 0.5
        k_, x_, ev_ = ut.dlqr(a_, b_, q_, r_)
        out.update({"a_" + tag: a_, "b_" + tag: b_, "q_" + tag: q_, "r_" + tag: r_, "k_" + tag: k_, "x_" + tag: x_,
                    "ev_" + tag: np.sort_complex(ev_)})
    _save("dlqr.npz", **out)

    # ------------------------------------------------------------------ 5. worked anchor of SURVEY 8c
    p = np.array([[0.1], [-0.2]]); Q = 0.2 * np.array([[.5, .2], [.2, .65]])
    k_ff = np.array([[0.3]]); k_fb = np.array([[0.4, -0.1]])
    mu = np.array([[0.05], [-0.02]]); s2 = np.array([[0.01], [0.04]])
    J = np.array([[0.1, 0.2, 0.3], [-0.1, 0.05, 0.2]])
    l = np.array([0.05, 0.02])
    const = lambda s, a: (mu, s2, J)
    a2 = np.array([[1, 0.05], [0, 0.98]]); b2 = np.array([[0.], [0.1]])
    pp, qp = gr.onestep_reachability(p, const, k_ff, l, l, None, None, 2.0, 0)
    pe, qe = gr.onestep_reachability(p, const, k_ff, l, l, Q, k_fb, 2.0, 0)
    pl, ql = gr.onestep_reachability(p, const, k_ff, l, l, Q, k_fb, 2.0, 0, a2, b2)
    h = np.vstack((np.eye(2), -np.eye(2)))
    dd = gr.lin_ellipsoid_safety_distance(pe, np.real(qe), h, np.ones((4, 1)), 1.0)
    _save("anchor.npz", p=p, Q=Q, k_ff=k_ff, k_fb=k_fb, mu=mu, var=s2, jac=J, l=l, a2=a2, b2=b2,
          p_point=pp, q_point=qp, p_ell=pe, q_ell=np.real(qe), p_lin=pl, q_lin=np.real(ql),
          d_safety=dd, dist=ue.distance_to_center(np.array([[0.2, 0.0]]), pe, np.real(qe)))


if __name__ == "__main__":
    main()
