"""CPU tests: the oracle (oracle/oracle_np.py) against every golden vector generated from the
reference itself (tests/golden/make_golden.py), plus independent cross-checks of the GP formulas."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle_np as orc


def _model(g):
    beta, inv_K, chol = orc.gp_fit(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])
    return dict(Z=g["Z"], beta=beta, inv_K=inv_K, chol=chol, lengthscale=g["lengthscale"],
                signal_var=g["signal_var"])


def test_ellipsoid_helpers_vs_reference():
    g = load_golden("ellipsoid.npz")
    for n in (2, 3, 4, 8):
        np.testing.assert_allclose(orc.ellipsoid_from_rectangle(g["rect_ub_%d" % n]), g["rect_q_%d" % n], rtol=1e-15)
        p, q = orc.sum_two_ellipsoids(g["sum_p1_%d" % n], g["sum_q1_%d" % n], g["sum_p2_%d" % n], g["sum_q2_%d" % n])
        np.testing.assert_allclose(p, g["sum_p_%d" % n], rtol=1e-15)
        np.testing.assert_allclose(q, g["sum_q_%d" % n], rtol=1e-14)
        _, qc = orc.sum_two_ellipsoids(g["sum_p1_%d" % n], g["sum_q1_%d" % n], g["sum_p2_%d" % n],
                                       g["sum_q2_%d" % n], c=0.7)
        np.testing.assert_allclose(qc, g["sumc_q_%d" % n], rtol=1e-14)
        d = orc.distance_to_center(g["dist_s_%d" % n], g["sum_p1_%d" % n], g["sum_q1_%d" % n])
        np.testing.assert_allclose(d, g["dist_d_%d" % n], rtol=1e-12)
    np.testing.assert_allclose(np.diag(g["known_rect"]), [0.03, 0.12, 0.27], rtol=1e-14)
    np.testing.assert_allclose(g["known_dist"], 1.0, rtol=1e-14)


def test_remainder_vs_reference():
    g = load_golden("remainder.npz")
    for tag in "1234":
        um, us = orc.compute_remainder_overapproximations(g["q_" + tag], g["k_fb_" + tag],
                                                          g["l_mu_" + tag], g["l_sigma_" + tag])
        np.testing.assert_allclose(um, g["u_mu_" + tag], rtol=1e-13)
        np.testing.assert_allclose(us, g["u_sigma_" + tag], rtol=1e-13)
    assert list(orc.sample_inside_polytope(g["poly_x"], g["poly_a"], g["poly_b"])) == [True, True, False]
    assert list(g["poly_res"]) == [True, True, False]


@pytest.mark.parametrize("name", ["gp_pend.npz", "gp_cart.npz"])
def test_gp_formulas_vs_reference_gp_pred(name):
    """fixtures hold the outputs of the reference's own _k_rbf/gp_pred evaluated on numbers."""
    g = load_golden(name)
    m = _model(g)
    np.testing.assert_allclose(m["beta"], g["beta"], rtol=1e-9, atol=1e-12)
    mu, var, jac = orc.gp_predict(g["x_new"], m["Z"], m["beta"], m["inv_K"], m["lengthscale"], m["signal_var"])
    np.testing.assert_allclose(mu, g["ref_mu"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, g["ref_var"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(orc.rbf_kernel(g["x_new"], m["Z"], g["signal_var"][0], g["lengthscale"][0]),
                               g["ref_kstar0"], rtol=1e-13)
    # explicit-inverse route == Cholesky route
    mu2, var2 = orc.gp_predict_chol(g["x_new"], m["Z"], m["beta"], m["chol"], m["lengthscale"], m["signal_var"])
    np.testing.assert_allclose(mu2, mu, rtol=1e-13)
    np.testing.assert_allclose(var2, var, rtol=0, atol=1e-10 * float(np.max(g["signal_var"])))


def test_gp_derivatives_vs_torch_autograd():
    """analytic d mu/dx, d var/dx, Hessian of mu vs torch-fp64 autograd."""
    g = load_golden("gp_pend.npz")
    m = _model(g)
    Z = torch.from_numpy(g["Z"])

    def post(x, d):
        ls = torch.from_numpy(g["lengthscale"][d])
        r2 = (((x[None, :] - Z) / ls) ** 2).sum(1)
        ks = g["signal_var"][d] * torch.exp(-0.5 * r2)
        mu = ks @ torch.from_numpy(m["beta"][:, d])
        var = g["signal_var"][d] - ks @ torch.from_numpy(m["inv_K"][d]) @ ks
        return mu, var

    x0 = g["x_new"][0]
    _, _, jac = orc.gp_predict(x0[None], m["Z"], m["beta"], m["inv_K"], m["lengthscale"], m["signal_var"])
    jv, hm = orc.gp_linearize_extras(x0, m["Z"], m["beta"], m["inv_K"], m["lengthscale"], m["signal_var"])
    for d in range(2):
        x = torch.from_numpy(x0.copy()).requires_grad_(True)
        mu, var = post(x, d)
        gmu, = torch.autograd.grad(mu, x, create_graph=True)
        gvar, = torch.autograd.grad(var, x, retain_graph=True)
        H = torch.stack([torch.autograd.grad(gmu[j], x, retain_graph=True)[0] for j in range(3)])
        np.testing.assert_allclose(jac[0, d], gmu.detach().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(jv[d], gvar.numpy(), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(hm[d], H.numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(jv, g["jac_var0"], rtol=1e-9, atol=1e-14)


def test_hand_cases_n1_n2():
    """N=1: mu = k y/(sf2+sn2), var = sf2 - k^2/(sf2+sn2)."""
    Z = np.array([[0.2, -0.1, 0.4]])
    Y = np.array([[0.7, -0.3]])
    ls = np.array([[0.5, 1.0, 2.0], [1.0, 1.0, 1.0]])
    sf2, sn2 = np.array([1.3, 0.6]), np.array([0.01, 0.02])
    beta, inv_K, _ = orc.gp_fit(Z, Y, ls, sf2, sn2)
    x = np.array([[0.0, 0.3, 0.1]])
    mu, var, jac = orc.gp_predict(x, Z, beta, inv_K, ls, sf2)
    for d in range(2):
        k = sf2[d] * np.exp(-0.5 * np.sum(((x - Z) / ls[d]) ** 2))
        den = sf2[d] + sn2[d] + orc.GPY_JITTER
        np.testing.assert_allclose(mu[0, d], k * Y[0, d] / den, rtol=1e-13)
        np.testing.assert_allclose(var[0, d], sf2[d] - k * k / den, rtol=1e-12)
        np.testing.assert_allclose(jac[0, d], k * Y[0, d] / den * (Z[0] - x[0]) / ls[d] ** 2, rtol=1e-12)


@pytest.mark.parametrize("name", ["reach_pend.npz", "reach_cart.npz", "reach_n3u2.npz"])
def test_reachability_vs_reference(name):
    """the oracle's ellipsoid algebra == the imported reference functions, same (mu,var,jac) fed."""
    g = load_golden(name)
    T, n_s = g["p"].shape
    n_u = g["k_ff"].shape[1]
    c = float(g["c_safety"])
    for tag, a, b in (("id", np.eye(n_s), np.zeros((n_s, n_u))), ("lin", g["a_lin"], g["b_lin"])):
        for t in range(T):
            p1, q1 = orc.onestep_reachability_from_gp(g["p"][t], None, g["k_ff"][t], None, g["mu"][t],
                                                      g["var"][t], g["jac"][t], g["l_mu"], g["l_sigma"], c, a, b)
            np.testing.assert_allclose(p1, g["p1_point_" + tag][t], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(q1, g["q1_point_" + tag][t], rtol=1e-13)
            p1, q1 = orc.onestep_reachability_from_gp(g["p"][t], g["Q"][t], g["k_ff"][t], g["k_fb"][t],
                                                      g["mu"][t], g["var"][t], g["jac"][t], g["l_mu"],
                                                      g["l_sigma"], c, a, b)
            np.testing.assert_allclose(p1, g["p1_ell_" + tag][t], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(q1, g["q1_ell_" + tag][t], rtol=1e-12, atol=1e-16)
    # full chain through the oracle's own GP (what the reference was driven with)
    m = _model(g)
    pa, qa = orc.multistep_reachability_batch(m, g["ms_p0"], g["ms_k_fb"], g["ms_k_ff"], g["l_mu"], g["l_sigma"],
                                              None, c, g["a_lin"], g["b_lin"], None)
    np.testing.assert_allclose(pa, g["ms_p_all"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(qa, g["ms_q_all"], rtol=1e-7, atol=1e-13)
    Tm = g["ms_p0"].shape[0]
    pa, qa = orc.multistep_reachability_batch(m, g["ms_p0"], g["ms_k_fb"], g["ms_k_ff"], g["l_mu"], g["l_sigma"],
                                              g["Q"][:Tm], c, g["a_lin"], g["b_lin"], g["k_fb"][:Tm])
    np.testing.assert_allclose(qa, g["ms_q_all_q0"], rtol=1e-7, atol=1e-13)
    for t in range(T):
        d = orc.lin_ellipsoid_safety_distance(g["p1_ell_id"][t][:, None], g["q1_ell_id"][t], g["h_mat"],
                                              g["h_vec"], c)
        np.testing.assert_allclose(d[:, 0], g["d_safety"][t], rtol=1e-13, atol=1e-15)
    # vectorised CPU baseline == per-query algebra
    vp, vq, _ = orc.onestep_reachability_vectorised(m, g["p"], g["Q"], g["k_ff"], g["k_fb"], g["l_mu"],
                                                    g["l_sigma"], c, g["a_lin"], g["b_lin"])
    np.testing.assert_allclose(vq, g["q1_ell_lin"], rtol=1e-8, atol=1e-14)


def test_anchor_numbers_of_the_survey():
    g = load_golden("anchor.npz")
    np.testing.assert_allclose(g["q_point"], np.diag([0.08, 0.32]), rtol=1e-14)
    np.testing.assert_allclose(g["q_ell"], [[0.605462201964151, 0.158620529130949],
                                            [0.158620529130949, 0.943186225924461]], rtol=1e-13)
    np.testing.assert_allclose(g["q_lin"], [[0.621617596208348, 0.17970909212829],
                                            [0.17970909212829, 0.933205766585243]], rtol=1e-13)
    np.testing.assert_allclose(g["dist"], 0.051607478675549, rtol=1e-12)
    a, b = np.eye(2), np.zeros((2, 1))
    p1, q1 = orc.onestep_reachability_from_gp(g["p"][:, 0], g["Q"], g["k_ff"][:, 0], g["k_fb"], g["mu"][:, 0],
                                              g["var"][:, 0], g["jac"], g["l"], g["l"], 2.0, a, b)
    np.testing.assert_allclose(q1, g["q_ell"], rtol=1e-13)
    um, us = orc.compute_remainder_overapproximations(g["Q"], g["k_fb"], g["l"], g["l"])
    np.testing.assert_allclose(um, [0.0080762036885, 0.0032304814754], rtol=1e-10)


@pytest.mark.parametrize("kt", ["mat52", "lin_rbf", "lin_mat52"])
def test_non_rbf_kernels_oracle_vs_reference(kt):
    """oracle kernels == the reference's _k_mat52/_k_lin/_k_lin_rbf/_k_lin_mat52 + gp_pred outputs."""
    g = load_golden("kern_%s.npz" % kt)
    hyp = []
    for d in range(2):
        pref = "hyp%d_" % d
        hyp.append({k[len(pref):]: g[k] for k in g.files if k.startswith(pref)})
    np.testing.assert_allclose(orc.kernel_matrix(kt, hyp[0], g["x_new"], g["Z"]), g["ref_kstar0"], rtol=1e-12, atol=1e-14)
    beta, inv_K = orc.gp_fit_k(g["Z"], g["Y"], [kt] * 2, hyp, g["noise_var"])
    mu, var = orc.gp_predict_k(g["x_new"], g["Z"], beta, inv_K, [kt] * 2, hyp)
    np.testing.assert_allclose(mu, g["ref_mu"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, g["ref_var"], rtol=0, atol=1e-9)
    # k(x,x) of the linear kernels is not constant
    kd = orc.kernel_diag(kt, hyp[0], g["x_new"])
    np.testing.assert_allclose(kd, np.diag(orc.kernel_matrix(kt, hyp[0], g["x_new"], g["x_new"])), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name", ["scen_invpend.npz", "scen_cartpole.npz"])
def test_reference_test_scenarios_on_reference_data(name):
    """canonical scenarios of test_gp_reachability_casadi.py:30-67 / test_safempc.py:56-69 on the
    reference's own data files; outputs produced by the imported reference functions."""
    g = load_golden(name)
    m = _model(g)
    n_s = g["p"].shape[0]
    n_u = g["k_ff"].shape[0]
    for tag, a, b in (("id", np.eye(n_s), np.zeros((n_s, n_u))), ("lin", g["a_lin"], g["b_lin"])):
        p1, q1, _ = orc.onestep_reachability_batch(m, g["p"].T, g["q0"][None], g["k_ff"].T, g["k_fb"][None],
                                                   g["L"], g["L"], 2.0, a, b)
        np.testing.assert_allclose(p1[0], g["p1_ell_" + tag][:, 0], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(q1[0], g["q1_ell_" + tag], rtol=1e-10)
        pa, qa = orc.multistep_reachability_batch(m, g["p"].T, g["k_fb_apply"][None], g["k_ff_all"][None], g["L"],
                                                  g["L"], None, 2.0, a, b, None)
        np.testing.assert_allclose(pa[0], g["ms_p_" + tag], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(qa[0], g["ms_q_" + tag], rtol=1e-8)


def test_oracle_vs_sklearn_gaussian_process():
    """Independent third-party witness for the exact-GP algebra (GPy itself is not installable here):
    scikit-learn's GaussianProcessRegressor with a fixed ConstantKernel * ARD-RBF kernel and
    alpha = noise + jitter must give the oracle's mean and variance."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    g = load_golden("gp_cart.npz")
    m = _model(g)
    mu, var, _ = orc.gp_predict(g["x_new"], m["Z"], m["beta"], m["inv_K"], m["lengthscale"], m["signal_var"])
    for d in range(g["Y"].shape[1]):
        kern = ConstantKernel(g["signal_var"][d], "fixed") * RBF(g["lengthscale"][d], "fixed")
        gpr = GaussianProcessRegressor(kern, alpha=g["noise_var"][d] + orc.GPY_JITTER, optimizer=None,
                                       normalize_y=False).fit(g["Z"], g["Y"][:, d])
        sm, ss = gpr.predict(g["x_new"], return_std=True)
        np.testing.assert_allclose(mu[:, d], sm, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(var[:, d], ss ** 2, rtol=0, atol=1e-9)


def test_oracle_mat52_vs_sklearn():
    """same witness for the Matern-5/2 kernel (sklearn Matern(nu=2.5) with ARD length scales)."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import Matern, ConstantKernel
    g = load_golden("kern_mat52.npz")
    hyp = []
    for d in range(2):
        pref = "hyp%d_" % d
        hyp.append({k[len(pref):]: g[k] for k in g.files if k.startswith(pref)})
    beta, inv_K = orc.gp_fit_k(g["Z"], g["Y"], ["mat52"] * 2, hyp, g["noise_var"])
    mu, var = orc.gp_predict_k(g["x_new"], g["Z"], beta, inv_K, ["mat52"] * 2, hyp)
    for d in range(2):
        kern = ConstantKernel(float(hyp[d]["variance"]), "fixed") * Matern(hyp[d]["lengthscale"], "fixed", nu=2.5)
        gpr = GaussianProcessRegressor(kern, alpha=g["noise_var"][d] + orc.GPY_JITTER, optimizer=None).fit(
            g["Z"], g["Y"][:, d])
        sm, ss = gpr.predict(g["x_new"], return_std=True)
        np.testing.assert_allclose(mu[:, d], sm, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(var[:, d], ss ** 2, rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["moments_pend.npz", "moments_cart.npz"])
def test_moment_propagation_vs_reference(name):
    """oracle's literal restatement == the reference's multi_step_taylor_symbolic / mean_equivalent_multistep
    evaluated on numbers (tests/golden/make_golden.py 4c)."""
    g = load_golden(name)
    m = _model(g)
    for tag, taylor in (("taylor", True), ("meaneq", False)):
        mu, sig = orc.multistep_moments_batch(m, g["mu0"], g["k_ff"], g["k_fb"], g["a_lin"], g["b_lin"], taylor)
        np.testing.assert_allclose(mu, g["mu_" + tag], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(sig, g["sigma_" + tag], rtol=1e-9, atol=1e-15)
    # the collapsed form the HIP kernel uses: Sigma_new = H Sigma H^T + diag(var)
    x = np.concatenate((g["mu_taylor"][0, 0], g["k_ff"][0, 1]))
    mg_, vg_, jg_ = orc._predict_one(m, x)
    n_s = g["mu0"].shape[1]
    K = g["k_fb"][0, 0]
    H = g["a_lin"] + jg_[:, :n_s] + (jg_[:, n_s:] + g["b_lin"]).dot(K)
    np.testing.assert_allclose(H.dot(g["sigma_taylor"][0, 0]).dot(H.T) + np.diag(vg_), g["sigma_taylor"][0, 1], rtol=1e-10)


@pytest.mark.parametrize("name", ["mc_pend.npz", "mc_cart.npz"])
def test_monte_carlo_propagation_vs_reference(name):
    """oracle's particle propagation == the reference's MonteCarloSafetyVerification.sample_n_step replaying the
    same standard-normal draws (tests/golden/make_golden.py 6); containment == its inside_ellipsoid_ratio."""
    g = load_golden(name)
    m = _model(g)
    S_all = orc.mc_sample_n_step(m, g["x0"], g["K"], g["k"], g["eps"])
    np.testing.assert_allclose(S_all, g["S_all"], rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(S_all[-1], g["S_last"])
    for i in range(S_all.shape[0]):
        inside = orc.distance_to_center(g["S_all"][i], g["ell_p"][i][:, None], g["ell_q"][i]) < 1.0
        np.testing.assert_array_equal(inside, g["inside"][i])
        assert abs(inside.mean() - g["ratio"][i]) < 1e-15


def test_information_gain_identity():
    """log det(I + K/s) == log det(K + s I) - N log s: the form the device computes from the factor."""
    syn = orc.make_synthetic(77, 40, 2, 1, 2)
    ig = orc.information_gain(syn["Z"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    for d in range(2):
        K = orc.rbf_kernel(syn["Z"], syn["Z"], syn["signal_var"][d], syn["lengthscale"][d])
        alt = np.linalg.slogdet(K + syn["noise_var"][d] * np.eye(40))[1] - 40 * np.log(syn["noise_var"][d])
        assert abs(ig[d] - alt) < 1e-9


@pytest.mark.parametrize("kt", ["rbf", "mat52", "lin_rbf", "lin_mat52"])
def test_second_order_outputs_of_all_kernels_vs_finite_differences(kt):
    """hand-differentiated d var/dx and Hessian of mu (oracle.gp_linearize_extras_k) against central differences
    of the oracle's own posterior; for rbf also against the dedicated closed form."""
    rng = np.random.default_rng(17)
    N, D = 40, 3
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, 2))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(2)]
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * 2, hyp, np.full(2, 1e-2))
    x = rng.uniform(-0.5, 0.5, D)
    jv, hm = orc.gp_linearize_extras_k(x, Z, beta, inv_K, [kt] * 2, hyp)
    eps = 1e-5
    for j in range(D):
        e = np.zeros(D)
        e[j] = eps
        vp = orc.gp_predict_k((x + e)[None], Z, beta, inv_K, [kt] * 2, hyp)[1][0]
        vm = orc.gp_predict_k((x - e)[None], Z, beta, inv_K, [kt] * 2, hyp)[1][0]
        np.testing.assert_allclose(jv[:, j], (vp - vm) / (2 * eps), rtol=1e-6, atol=1e-8)
        for d in range(2):
            gp_ = orc.kernel_derivatives(kt, hyp[d], x + e, Z)[1]
            gm_ = orc.kernel_derivatives(kt, hyp[d], x - e, Z)[1]
            np.testing.assert_allclose(hm[d][:, j], beta[:, d].dot(gp_ - gm_) / (2 * eps), rtol=1e-5,
                                       atol=1e-6 * np.abs(hm[d]).max())
    # the analytic gradient used above is itself the mean Jacobian (checked against differences of k)
    jac_fd = orc.gp_mean_jacobian_fd(x[None], Z, beta, [kt] * 2, hyp)[0]
    for d in range(2):
        np.testing.assert_allclose(beta[:, d].dot(orc.kernel_derivatives(kt, hyp[d], x, Z)[1]), jac_fd[d], rtol=1e-6,
                                   atol=1e-8)
    if kt == "rbf":
        ls = np.array([h["lengthscale"] for h in hyp])
        jv0, hm0 = orc.gp_linearize_extras(x, Z, beta, inv_K, ls, [h["variance"] for h in hyp])
        np.testing.assert_allclose(jv, jv0, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(hm, hm0, rtol=1e-12, atol=1e-13)


def _torch_kernel(kt, hyp, x, Z):
    """k(x, Z) (N,) in torch fp64 written directly from the reference's formulas
    (gp_models_utils_casadi.py:17-157: -2xy + x^2 + y^2 distances, product part on input dimension 1)."""
    import torch
    t = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64))

    def dist(a, b, ls):
        a, b = a / ls, b / ls
        r2 = -2.0 * (b * a[None, :]).sum(1) + (a * a).sum() + (b * b).sum(1)
        return torch.clamp(r2, min=0.0)

    def stationary(kind, a, b, var, ls):
        r2 = dist(a, b, ls)
        if kind == "rbf":
            return var * torch.exp(-0.5 * r2)
        r = torch.sqrt(r2)
        return var * (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * r2) * torch.exp(-np.sqrt(5.0) * r)

    D = Z.shape[1]
    if kt in ("rbf", "mat52"):
        return stationary(kt, x, t(Z), float(hyp["variance"]), t(hyp["lengthscale"]) * torch.ones(D, dtype=torch.float64))
    st = "rbf" if kt == "lin_rbf" else "mat52"
    vl = t(hyp["linear.variances"]) * torch.ones(D, dtype=torch.float64)
    vp = float(np.asarray(hyp["prod.linear.variances"]).reshape(-1)[0])
    x1, z1 = x[1:2], t(Z)[:, 1:2]
    k_st = stationary(st, x1, z1, float(hyp["prod.%s.variance" % st]), t(hyp["prod.%s.lengthscale" % st]).reshape(-1)[:1])
    return vp * x1[0] * z1[:, 0] * k_st + (t(Z) * vl[None, :] * x[None, :]).sum(1)


@pytest.mark.parametrize("kt", ["rbf", "mat52", "lin_rbf", "lin_mat52"])
def test_analytic_mean_jacobian_vs_torch_autograd(kt):
    """oracle.gp_mean_jacobian_k (what the GPU tests compare d mu/dx of the non-RBF kernels with, at rtol 1e-9)
    against torch-fp64 autograd of the reference's kernel formulas, plus the autograd Hessian of the mean."""
    import torch
    rng = np.random.default_rng({"rbf": 1, "mat52": 2, "lin_rbf": 3, "lin_mat52": 4}[kt])
    N, D = 50, 3
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, 2))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(2)]
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * 2, hyp, np.full(2, 1e-2))
    X = rng.uniform(-0.7, 0.7, (6, D))
    jac = orc.gp_mean_jacobian_k(X, Z, beta, [kt] * 2, hyp)
    for t_ in range(X.shape[0]):
        for d in range(2):
            f = lambda x: (_torch_kernel(kt, hyp[d], x, Z) * torch.from_numpy(beta[:, d])).sum()
            x = torch.tensor(X[t_], dtype=torch.float64, requires_grad=True)
            g_auto = torch.autograd.functional.jacobian(f, x).numpy()
            scale = np.abs(beta[:, d]).sum()
            np.testing.assert_allclose(jac[t_, d], g_auto, rtol=1e-10, atol=1e-12 * scale)
            if t_ == 0:
                h_auto = torch.autograd.functional.hessian(f, x).numpy()
                _, hm = orc.gp_linearize_extras_k(X[t_], Z, beta, inv_K, [kt] * 2, hyp)
                np.testing.assert_allclose(hm[d], h_auto, rtol=1e-9, atol=1e-11 * scale)


@pytest.mark.parametrize("kt", ["rbf", "mat52", "lin_rbf", "lin_mat52"])
def test_marginal_likelihood_gradient_vs_finite_differences(kt):
    """oracle.gp_nll_grad (the objective of train(opt_hyp=True)) against central differences in every
    hyper-parameter, and against scikit-learn's log marginal likelihood for the plain rbf case."""
    rng = np.random.default_rng(5)
    N, D = 30, 3
    Z = rng.uniform(-1, 1, (N, D))
    y = rng.standard_normal(N)
    hyp, nv = orc.make_hyp(kt, rng, D), 0.05
    nll, g = orc.gp_nll_grad(Z, y, kt, hyp, nv)
    eps = 1e-6
    for key, val in g.items():
        for j in range(val.size):
            def at(sign):
                h = {k: (np.array(v, dtype=float).copy() if np.ndim(v) > 0 else float(v)) for k, v in hyp.items()}
                n = nv
                if key == "noise_variance":
                    n = nv + sign * eps
                elif np.ndim(h[key]) == 0:
                    h[key] = h[key] + sign * eps
                else:
                    h[key].reshape(-1)[j] += sign * eps
                return orc.gp_nll_grad(Z, y, kt, h, n)[0]
            fd = (at(1) - at(-1)) / (2 * eps)
            assert abs(fd - val[j]) <= 2e-6 * max(1.0, abs(fd)), (key, j, fd, val[j])
    if kt == "rbf":
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import RBF, ConstantKernel
        kern = ConstantKernel(float(hyp["variance"]), "fixed") * RBF(hyp["lengthscale"], "fixed")
        gpr = GaussianProcessRegressor(kern, alpha=nv + orc.GPY_JITTER, optimizer=None).fit(Z, y)
        assert abs(-gpr.log_marginal_likelihood_value_ - nll) < 1e-9 * abs(nll)


@pytest.mark.parametrize("name,n_s,n_xin,n_u", [("reach_tz_cart.npz", 4, 3, 1), ("reach_tz_n3.npz", 3, 2, 2)])
def test_oracle_with_gp_input_transform_vs_reference_golden(name, n_s, n_xin, n_u):
    """the oracle's ellipsoid step fed with the GP outputs at t_z_gp @ state and the chain-ruled Jacobian reproduces
    what the reference's numeric functions return for the wrapped model (fixtures of make_golden.py, tz_case)."""
    g = load_golden(name)
    tz = g["tz"]
    beta, inv_K, _ = orc.gp_fit(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])
    model = dict(Z=g["Z"], beta=beta, inv_K=inv_K, lengthscale=g["lengthscale"], signal_var=g["signal_var"])
    for t in range(g["p"].shape[0]):
        z = np.hstack((tz.dot(g["p"][t]), g["k_ff"][t]))
        mu, var, jac = orc._predict_one(model, z)
        jac = np.hstack((jac[:, :n_xin].dot(tz), jac[:, n_xin:]))
        p1, q1 = orc.onestep_reachability_from_gp(g["p"][t], g["Q"][t], g["k_ff"][t], g["k_fb"][t], mu, var, jac,
                                                  g["l_mu"], g["l_sigma"], float(g["c_safety"]), g["a_lin"], g["b_lin"])
        np.testing.assert_allclose(p1, g["p1_ell"][t], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(q1, g["q1_ell"][t], rtol=1e-10, atol=1e-16)
