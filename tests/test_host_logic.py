"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
Python mirror keeps the reference's surface and error behaviour, host helpers match the golden
vectors, and nothing in the product package routes through the oracle."""
import os
import re

import numpy as np
import pytest

from conftest import load_golden, ROOT


def test_library_exports_every_declared_symbol(lib_built):
    from safe_exploration_amd import _lib
    header = open(os.path.join(ROOT, "include", "safereach.h")).read()
    declared = set(re.findall(r"\b(sr_[a-z_0-9]+)\s*\(", header))
    declared -= {"sr_gp"}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), "libsafereach.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert set(_lib.SIGNATURES) == declared
    assert _lib.lib.sr_version() >= 100


def test_comm_library_exports_every_declared_symbol(lib_built):
    """include/safereach_comm.h <-> libsafereach_comm.so (RCCL replication for torch-free hosts); load only, no calls"""
    import ctypes
    header = open(os.path.join(ROOT, "include", "safereach_comm.h")).read()
    declared = set(re.findall(r"\b(sr_comm_[a-z_0-9]+)\s*\(", header))
    assert declared == {"sr_comm_init_all", "sr_comm_bcast", "sr_comm_destroy", "sr_comm_last_error"}
    from safe_exploration_amd import _lib  # noqa: F401  (loads torch's HIP runtime and libsafereach.so first)
    path = os.path.join(ROOT, "safe_exploration_amd", "libsafereach_comm.so")
    assert os.path.exists(path), "make -C safe_exploration_amd/csrc builds it"
    try:
        comm = ctypes.CDLL(path)
    except OSError as exc:
        pytest.skip("librccl.so not loadable here: %s" % exc)
    for name in sorted(declared):
        assert hasattr(comm, name), "libsafereach_comm.so does not export %s" % name


def test_wait_flag_gives_up_and_gives_the_core_away(lib_built):
    """sr_wait_flag is pure host code: a sequence number that never arrives ends in SR_ESTATE after the time-out (not
    5 s of spinning whatever the caller asked for), and a waiting call leaves the core to others after the first 5 ms
    (process CPU time well below wall time); a number that is already there returns at once."""
    import ctypes
    import time
    from safe_exploration_amd import _lib
    flag = ctypes.c_ulonglong(7)
    assert _lib.lib.sr_wait_flag(ctypes.byref(flag), 7, 1.0) == 0
    c0, w0 = time.process_time(), time.perf_counter()
    rc = _lib.lib.sr_wait_flag(ctypes.byref(flag), 8, 0.3)
    cpu, wall = time.process_time() - c0, time.perf_counter() - w0
    assert rc == -4 and b"not seen" in _lib.lib.sr_last_error()
    assert 0.3 <= wall < 1.0
    assert cpu < 0.6 * wall, "busy-waited %.3f of %.3f s" % (cpu, wall)


def test_no_gpu_fails_loudly(lib_built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, _lib
    assert _lib.device_count() == 0
    gp = SimpleGPModel(2, 2, 1)
    with pytest.raises(RuntimeError):
        gp.train(np.zeros((4, 3)), np.zeros((4, 2)), opt_hyp=False)
    with pytest.raises(RuntimeError):
        gp.predict(np.zeros((1, 3)))
    with pytest.raises(RuntimeError):
        reach.ellipsoid_step_batch(np.zeros((1, 2)), np.zeros((1, 1)), np.zeros((1, 2)), np.ones((1, 2)),
                                   None, np.ones(2), np.ones(2))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "safe_exploration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), "%s mentions the oracle" % f
    # bench.py: the oracle is imported only inside the cpu_baseline* legs (after the timed region)
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    inside = 0
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for n in ast.walk(fn):
            if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n):
                assert fn.name.startswith("cpu_baseline"), "bench.py imports the oracle inside %s()" % fn.name
                inside += 1
    total = sum(1 for n in ast.walk(tree) if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n))
    assert inside == total >= 1


def test_simple_gp_model_surface(lib_built):
    from safe_exploration_amd import SimpleGPModel, StateSpaceModel
    gp = SimpleGPModel(2, 2, 1)
    assert isinstance(gp, StateSpaceModel)
    assert (gp.n_s_out, gp.n_s_in, gp.n_u, gp.num_states, gp.num_actions) == (2, 2, 1, 2, 1)
    assert gp.has_jacobian and gp.has_reverse and not gp.gp_trained      # reverse mode is implemented
    assert gp.kern_types == ["rbf", "rbf"] and gp.beta is None and gp.inv_K is None
    assert set(gp.hyp[0]) == {"lengthscale", "variance"} and gp.hyp[0]["lengthscale"].shape == (3,)
    with pytest.raises(ValueError):
        SimpleGPModel(2, 2, 1, kern_types=["rbf", "nope"])
    g = SimpleGPModel(2, 2, 1, kern_types=["mat52", "lin_mat52"])
    assert set(g.hyp[1]) == {"prod.mat52.lengthscale", "prod.mat52.variance", "prod.linear.variances",
                             "linear.variances"}
    kp = g._pack_kernel_params()
    assert kp.shape == (2, 12) and kp[0, 0] == 1.0 and kp[0, 2] == 1.0 and kp[1, 2] == 0.0
    assert kp[1, 3 + 1] == 1.0 and kp[1, 3 + 0] == 0.0 and kp[1, 3 + 3 + 1] == 1.0     # product part on dim 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # opt_hyp=True evaluates the likelihood on the GPU
        gp.train(np.zeros((4, 3)), np.zeros((4, 2)))
    assert gp._free_hyp(0) == [("lengthscale", 3), ("variance", 1), ("noise_variance", 1)]
    fixed = SimpleGPModel(2, 2, 1, hyp=[{"lengthscale": np.ones(3), "noise_variance": 0.1}] * 2)
    assert fixed._free_hyp(1) == [("variance", 1)]
    fixed._set_free(1, np.array([2.5]))
    assert fixed.hyp[1]["variance"] == 2.5 and fixed._get_free(1).tolist() == [2.5]
    with pytest.raises(ValueError):
        gp.train(np.zeros((4, 2)), np.zeros((4, 2)), opt_hyp=False)
    with pytest.raises(RuntimeError):
        gp.predict_device(np.zeros((1, 3)))
    d = gp.to_dict()
    assert set(d) == {"x", "y", "kern_types", "hyp", "beta", "inv_K"}
    g2 = SimpleGPModel.from_dict({"n_s_in": 2, "n_s_out": 2, "n_u": 1, "x": np.zeros((3, 3)),
                                  "y": np.zeros((3, 2)), "train": False})
    assert not g2.gp_trained
    import copy
    g3 = copy.deepcopy(gp)
    assert g3 is not gp and g3.hyp is gp.hyp


def test_state_space_model_abstract_surface():
    from safe_exploration_amd import StateSpaceModel
    ssm = StateSpaceModel(3, 2)
    for call in (lambda: ssm.predict(None, None), lambda: ssm(None, None),
                 lambda: ssm.linearize_predict(None, None), lambda: ssm.get_reverse(None),
                 lambda: ssm.get_linearize_reverse(None), lambda: ssm.update_model(None, None)):
        with pytest.raises(NotImplementedError):
            call()
    with pytest.raises(ImportError):       # casadi is an optional dependency of the caller
        ssm.get_forward_model_casadi()


def test_utils_ellipsoid_host_helpers_vs_reference_golden():
    from safe_exploration_amd import utils_ellipsoid as ue
    g = load_golden("ellipsoid.npz")
    for n in (2, 3, 4, 8):
        np.testing.assert_allclose(ue.ellipsoid_from_rectangle(g["rect_ub_%d" % n]), g["rect_q_%d" % n], rtol=1e-15)
        p, q = ue.sum_two_ellipsoids(g["sum_p1_%d" % n], g["sum_q1_%d" % n], g["sum_p2_%d" % n], g["sum_q2_%d" % n])
        np.testing.assert_allclose(q, g["sum_q_%d" % n], rtol=1e-14)
        np.testing.assert_allclose(p, g["sum_p_%d" % n], rtol=1e-15)
        for k in (3, 4):
            pn, qn = ue.sum_ellipsoids(g["msum_p_in_%d_%d" % (k, n)], g["msum_q_in_%d_%d" % (k, n)],
                                       g["msum_l_%d_%d" % (k, n)])
            np.testing.assert_allclose(pn, g["msum_p_%d_%d" % (k, n)], rtol=1e-14)
            np.testing.assert_allclose(qn, g["msum_q_%d_%d" % (k, n)], rtol=1e-13)
        d = ue.distance_to_center(g["dist_s_%d" % n], g["sum_p1_%d" % n], g["sum_q1_%d" % n])
        np.testing.assert_allclose(d, g["dist_d_%d" % n], rtol=1e-12)
        assert list(ue.sample_inside_ellipsoid(g["dist_s_%d" % n], g["sum_p1_%d" % n], g["sum_q1_%d" % n], 3.0)) \
            == list(g["inside_%d" % n])
    # the reference test's geometric property: box corners lie on the ellipsoid (test_utils_ellipsoid.py:13-25)
    ub = np.array([0.1, 0.2, 0.3])
    q = ue.ellipsoid_from_rectangle(ub)
    corners = np.array([[sx * ub[0], sy * ub[1], sz * ub[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    np.testing.assert_allclose(ue.distance_to_center(corners, np.zeros((3, 1)), q), 1.0, rtol=1e-12)
    with pytest.raises(AssertionError):
        ue.ellipsoid_from_rectangle(np.array([0.1, 0.0]))
    with pytest.raises(AssertionError):
        ue.sum_ellipsoids(np.zeros((1, 2)), np.eye(2)[None])


def test_utils_host_helpers():
    from safe_exploration_amd import utils
    g = load_golden("remainder.npz")
    assert list(utils.sample_inside_polytope(g["poly_x"], g["poly_a"], g["poly_b"])) == [True, True, False]
    k_fb, p, x, k_ff = np.array([[1.0, 2.0]]), np.array([[0.1], [0.2]]), np.array([[0.3], [0.1]]), np.array([[0.5]])
    np.testing.assert_allclose(utils.feedback_ctrl(x, k_ff, k_fb, p), [[0.5]])
    assert utils.feedback_ctrl(x, k_ff) is k_ff
    assert utils.array_of_vec_to_array_of_mat(np.arange(12.0).reshape(2, 6), 2, 3).shape == (2, 2, 3)


def test_safety_distance_shape_asserts(lib_built):
    from safe_exploration_amd import gp_reachability as reach
    with pytest.raises(AssertionError):
        reach.lin_ellipsoid_safety_distance(np.zeros((2,)), np.eye(2), np.eye(2), np.ones((2, 1)))
    with pytest.raises(AssertionError):
        reach.lin_ellipsoid_safety_distance(np.zeros((2, 1)), np.eye(3), np.eye(2), np.ones((2, 1)))


def test_workload_generator_shapes():
    from safe_exploration_amd import workload
    prob = workload.make_problem(1, 50, 2, 1, 16)
    assert prob["Z"].shape == (50, 3) and prob["Q"].shape == (16, 2, 2)
    assert np.all(np.linalg.eigvalsh(prob["Q"]) > 0)
    r = workload.random_rollout_controls(3, 8, 15, 4, 1)
    assert r["k_fb"].shape == (8, 14, 1, 4) and r["k_ff"].shape == (8, 15, 1) and r["p0"].shape == (8, 4)
    assert abs(np.std(r["k_fb"]) - 0.1) < 0.02       # uncertainty_propagation_runner.py:32-33


def test_reshape_derivatives():
    from safe_exploration_amd import utils
    d3 = np.arange(2 * 3 * 4, dtype=float).reshape(2, 3, 4)
    d2 = utils.reshape_derivatives_3d_to_2d(d3)
    assert d2.shape == (6, 4) and d2[4, 1] == d3[1, 1, 1]


def test_true_system_rollouts_vs_reference_golden():
    """simulate_trajectory / verify_trajectory_safety / trajectory_inside_ellipsoid (gp_reachability.py:253-356)
    against the reference's own functions driven with the same stand-in environment (make_golden.py 7)."""
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden("traj.npz")

    class ToyEnv(object):
        n_s, n_u = 2, 1

        def simulate_onestep(self, x, u):
            x, u = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(u, dtype=np.float64).reshape(-1)
            return g["A"].dot(x) + g["Bm"].dot(u) + 0.05 * np.sin(x), None

    env, n = ToyEnv(), g["k_ff"].shape[0]
    x_all = reach.simulate_trajectory(env, g["p_0"], g["k_fb"], g["k_ff"], g["p_all"][:n - 1])
    np.testing.assert_allclose(x_all, g["x_all"], rtol=1e-13, atol=1e-15)
    inside = reach.trajectory_inside_ellipsoid(env, g["p_0"], g["p_all"], g["q_all"], g["k_fb"], g["k_ff"])
    np.testing.assert_array_equal(inside, g["inside"])
    assert inside.dtype == bool and not inside.all() and inside.any()
    wide = np.full((4, 1), 5.0)
    ok, xa = reach.verify_trajectory_safety(env, g["p_0"], g["k_fb"], g["k_ff"], g["p_all"][:n - 1], g["h_mat"], wide,
                                            g["h_mat"], wide)
    assert ok == bool(g["ok_wide"]) and ok is True
    np.testing.assert_array_equal(xa, x_all)
    ok2, _ = reach.verify_trajectory_safety(env, g["p_0"], g["k_fb"], g["k_ff"], g["p_all"][:n - 1], g["h_mat"],
                                            np.full((4, 1), float(g["lim"])))
    assert ok2 == bool(g["ok_tight"]) and ok2 is False
    # n = 1: no feedback stage at all
    x1 = reach.simulate_trajectory(env, g["p_0"], None, g["k_ff"][:1], None)
    np.testing.assert_allclose(x1[1], g["x_all"][1], rtol=1e-13)


STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standin")


def _linear_ssm():
    from safe_exploration_amd import StateSpaceModel

    class Linear(StateSpaceModel):
        """mu = A z + 0.5 (c.z)^2 e_0, var = 0.1 + (w.z)^2 (z = [x; u]): closed forms for every derivative the
        evaluator asks for, reverse mode included."""
        A = np.array([[0.9, 0.1, 0.3], [-0.2, 0.8, 0.5]])
        w = np.array([0.3, -0.4, 0.2])
        c = np.array([0.5, 0.25, -1.0])

        def _all(self, z):
            mu = self.A.dot(z)
            mu[0] += 0.5 * self.c.dot(z) ** 2
            jm = self.A.copy()
            jm[0] += self.c.dot(z) * self.c
            hm = np.zeros((2, 3, 3))
            hm[0] = np.outer(self.c, self.c)
            return (mu[:, None], np.full((2, 1), 0.1 + self.w.dot(z) ** 2), jm,
                    np.tile(2 * self.w.dot(z) * self.w, (2, 1)), hm)

        def linearize_predict(self, states, actions, jacobians=False, full_cov=False):
            out = self._all(np.hstack((np.asarray(states), np.asarray(actions)))[0])
            self._linearize_forward_cache = out[2:]
            return out if jacobians else out[:3]

        def get_linearize_reverse(self, seed):
            jm, jv, hm = self._linearize_forward_cache
            seed = np.asarray(seed, dtype=np.float64).reshape(-1)
            g = seed[:2].dot(jm) + seed[2:4].dot(jv) + np.einsum("ij,ijk->k", seed[4:].reshape(2, 3), hm)
            return g[:2, None], g[2:, None]

    return Linear


def _drive_evaluator(ev, ssm_cls, has_reverse):
    """the call sequence IPOPT drives through casadi: forward, Jacobian callback, reverse callback"""
    z = np.array([0.2, -0.1, 0.4])
    x, u = z[:2, None], z[2:, None]
    ref = ssm_cls(2, 1)._all(z)
    assert ev.get_n_in() == 2 and ev.get_n_out() == 3 and ev.get_sparsity_out(2) == (2, 3)
    mu, var, jac = ev.eval([x, u])
    np.testing.assert_allclose(mu, ref[0])
    np.testing.assert_allclose(var, ref[1])
    np.testing.assert_allclose(jac, ref[2])
    jfun = ev.get_jacobian("jac", [], [], {})
    assert jfun.get_n_in() == 5 and jfun.get_n_out() == 1 and jfun.get_sparsity_out(0) == (2 * 2 + 2 * 3, 3)
    if hasattr(ev, "jac_mu_order"):
        # this package's evaluator numbers the d jac_mean / dz rows by CasADi's vec rule (column-major) by default;
        # "C" gives the rows of the reference's helper (utils.py:357-380), which the comparisons below use
        assert ev.jac_mu_order == "F"
        (stacked_f,) = jfun.eval([x, u, mu, var, jac])
        np.testing.assert_allclose(np.array(stacked_f)[4:], np.transpose(ref[4], (1, 0, 2)).reshape(6, 3))
        ev.jac_mu_order = "C"
    (stacked,) = jfun.eval([x, u, mu, var, jac])
    stacked = np.array(stacked)
    assert stacked.shape == (2 * 2 + 2 * 3, 3)                 # [jac_mu; jac_var; d jac_mu / dz]
    np.testing.assert_allclose(stacked[:2], ref[2])
    np.testing.assert_allclose(stacked[2:4], ref[3])
    np.testing.assert_allclose(stacked[4:], ref[4].reshape(6, 3))
    assert ev.has_jacobian() and ev.has_reverse(1) == has_reverse and not ev.has_reverse(2) and not ev.has_forward(1)
    if not has_reverse:
        with pytest.raises(ValueError):
            ev.get_reverse(1, "rev", [], [], {})
        return None
    bfun = ev.get_reverse(1, "rev", [], [], {})
    assert bfun.get_n_in() == 8 and bfun.get_n_out() == 2 and bfun.get_sparsity_in(7) == (2, 3)
    return bfun, stacked, (x, u, mu, var, jac)


def test_casadi_evaluator_contract_forward_jacobian_reverse(monkeypatch):
    """This package's CasadiSSMEvaluator (state_space_models.py:214-566 of the reference) on a stand-in casadi
    module: arities, sparsities, the stacked (2n + nD) x D Jacobian and the reverse-mode adjoints -- the three
    callbacks IPOPT reaches -- against closed forms.  ``get_forward_model_casadi`` passes has_jacobian / has_reverse
    of the model like state_space_models.py:166."""
    import sys
    monkeypatch.syspath_prepend(STANDIN)
    monkeypatch.delitem(sys.modules, "casadi", raising=False)
    import casadi
    Linear = _linear_ssm()
    ssm = Linear(2, 1, has_jacobian=True, has_reverse=True)
    ev = ssm.get_forward_model_casadi(True)
    assert type(ev).__name__ == "CasadiSSMEvaluator" and ev.ssm is not ssm and ev.constructed == "CasadiModelEvaluator"
    assert ev.v_has_reverse is True and ev.v_has_jacobian is True
    bfun, stacked, (x, u, mu, var, jac) = _drive_evaluator(ev, Linear, True)
    rng = np.random.default_rng(0)
    s_mu, s_var, s_jac = rng.standard_normal((2, 1)), rng.standard_normal((2, 1)), rng.standard_normal((2, 3))
    adj_x, adj_u = bfun.eval([casadi.DM(a) for a in (x, u, mu, var, jac, s_mu, s_var, s_jac)])
    want = np.concatenate((s_mu.ravel(), s_var.ravel(), s_jac.ravel())).dot(stacked)     # seed^T J, row-major jac seed
    np.testing.assert_allclose(np.array(adj_x).ravel(), want[:2], rtol=1e-13)
    np.testing.assert_allclose(np.array(adj_u).ravel(), want[2:], rtol=1e-13)
    # numeric call path with shape checks against the declared sparsities
    out = ev(x, u)
    assert [o.shape for o in out] == [(2, 1), (2, 1), (2, 3)]
    # a model without reverse mode: the flag travels, get_reverse refuses
    ev2 = Linear(2, 1).get_forward_model_casadi(True)
    assert ev2.v_has_reverse is False
    _drive_evaluator(ev2, Linear, False)
    with pytest.raises(ValueError):
        from safe_exploration_amd import state_space_models as ssm_mod
        ssm_mod.CasadiSSMEvaluator(ssm, True, False, False)
    # not linearised: two outputs, (2n) x D Jacobian
    from safe_exploration_amd import state_space_models as ssm_mod
    assert ssm_mod.CasadiSSMEvaluator is type(ev)


@pytest.mark.skipif(not os.path.isdir("/root/reference/safe_exploration"),
                    reason="needs the reference checkout (build container only)")
def test_reference_casadi_evaluator_runs_on_this_surface(monkeypatch):
    """The REFERENCE's own CasadiSSMEvaluator (state_space_models.py:214-566) driven over this package's
    StateSpaceModel surface with the same stand-in casadi module, through JacFun.eval and BackFun.eval
    (:384-417, :534-562): it must accept the surface unchanged and produce what this package's evaluator does."""
    import sys
    monkeypatch.syspath_prepend(STANDIN)
    monkeypatch.syspath_prepend("/root/reference")
    for m in [k for k in sys.modules if k == "casadi" or k.startswith("safe_exploration.") or k == "safe_exploration"]:
        monkeypatch.delitem(sys.modules, m)
    import casadi
    from safe_exploration.state_space_models import CasadiSSMEvaluator as RefEvaluator
    Linear = _linear_ssm()
    ssm = Linear(2, 1, has_jacobian=True, has_reverse=True)
    ev = RefEvaluator(ssm, True, ssm.has_jacobian, ssm.has_reverse)
    bfun, stacked, (x, u, mu, var, jac) = _drive_evaluator(ev, Linear, True)
    mine = ssm.get_forward_model_casadi(True)
    mine.jac_mu_order = "C"                      # the reference helper's row order
    (stacked_mine,) = mine.get_jacobian("jac", [], [], {}).eval([x, u, mu, var, jac])
    np.testing.assert_array_equal(stacked, np.array(stacked_mine))
    # reverse: the reference flattens the jac_mean seed with casadi's column-major reshape (:555-556) while its own
    # GPyTorchSSM caches jac_mean row-major (ssm_pytorch/gaussian_process.py:372); the two agree for a seed whose
    # two flattenings coincide, e.g. one that only touches the first column
    s_mu, s_var = np.array([[0.3], [-0.7]]), np.array([[1.1], [0.2]])
    s_jac = np.zeros((2, 3))
    ssm.linearize_predict(x.T, u.T, True)
    adj_ref = bfun.eval([casadi.DM(a) for a in (x, u, mu, var, jac, s_mu, s_var, s_jac)])
    adj_mine = mine.get_reverse(1, "rev", [], [], {}).eval([casadi.DM(a) for a in (x, u, mu, var, jac, s_mu, s_var, s_jac)])
    for a, b in zip(adj_ref, adj_mine):
        np.testing.assert_allclose(np.array(a), np.array(b), rtol=1e-6)      # the reference casts its seed to float32


def test_dlqr_and_name_aliases_vs_reference_golden():
    """utils.dlqr (utils.py:20-35) against the reference's own gains; the reference's module / function names of the
    moment propagation resolve to the batched numeric implementation."""
    from safe_exploration_amd import utils
    g = load_golden("dlqr.npz")
    for tag in "ab":
        k, x, ev = utils.dlqr(g["a_" + tag], g["b_" + tag], g["q_" + tag], g["r_" + tag])
        np.testing.assert_allclose(k, g["k_" + tag], rtol=1e-10)
        np.testing.assert_allclose(x, g["x_" + tag], rtol=1e-10)
        np.testing.assert_allclose(np.sort_complex(ev), g["ev_" + tag], rtol=1e-9, atol=1e-12)
        assert np.abs(ev).max() < 1.0
    # the reference's module and function names (uncertainty_propagation_casadi.py:11-283) are importable as they are
    from safe_exploration_amd import uncertainty_propagation_casadi as upc
    assert upc.multi_step_taylor_symbolic is upc.multi_step_taylor
    for name in ("one_step_taylor", "mean_equivalent_multistep", "one_step_mean_equivalent"):
        assert callable(getattr(upc, name))


def test_wait_flag_is_host_code_and_times_out():
    """sr_wait_flag is the host half of the completion mailbox: it returns at once when the sequence number is there,
    sees a store from another thread, and gives up after its timeout instead of hanging a CasADi callback."""
    import ctypes
    import threading
    import time
    from safe_exploration_amd._lib import lib, SR_OK
    flag = (ctypes.c_ulonglong * 1)(41)
    assert lib.sr_wait_flag(ctypes.cast(flag, ctypes.c_void_p), 41, 0.01) == SR_OK
    t0 = time.perf_counter()
    assert lib.sr_wait_flag(ctypes.cast(flag, ctypes.c_void_p), 42, 0.05) != SR_OK
    assert 0.04 < time.perf_counter() - t0 < 2.0
    assert b"42" in lib.sr_last_error()

    def later():
        time.sleep(0.02)
        flag[0] = 42
    th = threading.Thread(target=later)
    th.start()
    assert lib.sr_wait_flag(ctypes.cast(flag, ctypes.c_void_p), 42, 5.0) == SR_OK
    th.join()


def test_only_the_documented_kernels_use_scratch():
    """Register spills are a silent cost (a branch around one prologue put sr_stream_mfma_kernel<2> into scratch: 50 -> 75 us
    per launch at N = 5000, and only a latency table showed it).  The kernels that still spill are the ones DESIGN.md and
    include/safereach.h name; every other kernel of the library compiles scratch-free for gfx950."""
    import shutil
    import sys
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    rows = kernel_resources.collect()
    assert len(rows) > 200
    allowed = {"void sr_ellipsoid_kernel<8, 3>", "void sr_ellipsoid_kernel<8, 4>",
               "void sr_gp_small_general_kernel<256, 8>", "void sr_gp_small_general_kernel<384, 8>",
               "void sr_gp_small_general_kernel<512, 8>",
               # the opt-in tile-flow Cholesky (csrc/sr_flow.hip): both tile sizes in one resident kernel, 32 bytes per lane;
               # its resident diagonal-block workgroup (the launched kernel's block routine inside a loop over the blocks, at
               # 128 VGPRs for 16 wavefronts): 44 bytes per lane
               "sr_flow_worker_kernel", "sr_flow_diag_server_kernel"}
    spilled = {r["kernel"] for r in rows if int(r.get("ScratchSize", 0)) > 0}
    assert spilled <= allowed, sorted(spilled - allowed)


def test_bench_rank_section_and_single_rank_gather():
    """The self-diagnosis of an N > 1 bench line, host side: min / max / slowest / spread of the ranks' own
    times, and the one-rank form of the gather (no process group needed)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rk = bench.rank_section([45.1, 45.9, 44.8, 45.0])
    assert rk["ms_per_step"] == [45.1, 45.9, 44.8, 45.0]
    assert rk["ms_per_step_min"] == 44.8 and rk["ms_per_step_max"] == 45.9 and rk["slowest_rank"] == 1
    assert abs(rk["spread"] - (45.9 - 44.8) / 44.8) < 1e-15
    one = bench.gather_rank_stats([1.5, 2, 3.25], "cpu", 1)
    assert one.shape == (1, 3) and list(one[0]) == [1.5, 2.0, 3.25]
    assert bench.XGMI_LINK_GBS == 153.0
    args = bench.parse_args(["--workload", "c4", "--gpus", "2"])
    assert args.c4_replicas is False and bench.parse_args(["--c4-replicas"]).c4_replicas is True


def _flow_tasks(nb, band, P, segs):
    """the tile-flow Cholesky's task order as csrc/sr_flow.hip decodes it (fl_decode and the position -> task step of the
    worker kernel), in Python: a list of (kind, ...) tuples in the order the positions are handed out"""
    near = lambda i: min(nb - 1 - i, band)                                           # noqa: E731
    upd_rows = lambda q, bi: max(0, nb - bi - band - 1) if bi < (q + 2) * P else nb - bi   # noqa: E731

    def critical(qi):
        s = max(i for i in range(nb) if segs[i][0] <= qi)
        e = qi - segs[s][0]
        n_sol = 4 * near(s)
        if e < n_sol:
            return ("solve64", s, s + 1 + (e >> 2), (e >> 1) & 1, e & 1)
        e -= n_sol
        i = s + 1
        p = i // P
        k_lo, apt = ((p - 1) * P, p - 1) if p >= 1 else (0, 0)
        if e < 3:
            return ("upd64", i, i, 1 if e == 2 else 0, 0 if e == 0 else 1, k_lo, apt)
        e -= 3
        return ("upd64", i, i + 1 + (e >> 2), (e >> 1) & 1, e & 1, k_lo, apt)

    def far(qi):
        i = max(r for r in range(nb) if segs[r][1] <= qi and (segs[r + 1][1] > qi or segs[r + 1][1] == segs[r][1] and False))
        return ("far", i, i + 1 + near(i) + (qi - segs[i][1]), (i // P) * P, i // P)

    def update(qi):
        q, base = 0, 0
        while True:
            cnt = sum(upd_rows(q, bi) for bi in range((q + 1) * P, nb))
            if qi < base + cnt:
                break
            base += cnt
            q += 1
        idx, bi = qi - base, (q + 1) * P
        while idx >= upd_rows(q, bi):
            idx -= upd_rows(q, bi)
            bi += 1
        return ("UPD", q, bi, (bi + band + 1 if bi < (q + 2) * P else bi) + idx)

    out = []
    for i in range(nb):
        nbk, nc, nf = segs[i + 1][2] - segs[i][2], segs[i + 1][0] - segs[i][0], segs[i + 1][1] - segs[i][1]
        n_sol = 4 * near(i)
        out += [update(segs[i][2] + e) for e in range(nbk)]
        out += [critical(segs[i][0] + e) for e in range(n_sol)]
        out += [far(segs[i][1] + e) for e in range(nf)]
        out += [critical(segs[i][0] + e) for e in range(n_sol, nc)]
    return out


@pytest.mark.parametrize("nb,band,panel", [(3, 2, 2), (5, 3, 4), (7, 2, 2), (16, 2, 3), (23, 1, 3), (40, 2, 4), (12, 2, 100), (9, 0, 2)])
def test_tile_flow_plan_covers_every_block_once_and_orders_dependencies(lib_built, nb, band, panel):
    """The task plan of the tile-flow Cholesky (csrc/sr_flow.h, sr_flow_plan; the decoding mirrored above).  Every tile of the
    factor is produced exactly once, every block takes every panel in front of it exactly once (by a UPD task, or left-looking
    inside its band task), and whatever a task waits for stands IN FRONT of it in the one order the positions are handed out
    in -- which is what makes the resident kernel deadlock-free for any number of resident workgroups."""
    import ctypes
    from safe_exploration_amd import _lib
    sg = (ctypes.c_int * (4 * (nb + 1)))()
    tot = (ctypes.c_long * 4)()
    assert _lib.lib.sr_test_flow_plan(nb, band, panel, sg, tot) == 0
    segs = [tuple(sg[4 * i:4 * i + 4]) for i in range(nb + 1)]
    assert [s[3] for s in segs] == sorted(s[3] for s in segs) and segs[nb][3] == tot[3] == tot[0] + tot[1] + tot[2]
    tasks = _flow_tasks(nb, band, panel, segs)
    assert len(tasks) == tot[3] and len(set(tasks)) == len(tasks)
    pos = {t: k for k, t in enumerate(tasks)}
    P = panel
    near = lambda i: min(nb - 1 - i, band)                                           # noqa: E731
    # -- coverage: factor tiles
    solved64 = {(t[1], t[2], t[3], t[4]) for t in tasks if t[0] == "solve64"}
    assert solved64 == {(i, j, rh, ch) for i in range(nb) for j in range(i + 1, i + 1 + near(i)) for rh in (0, 1) for ch in (0, 1)}
    assert {(t[1], t[2]) for t in tasks if t[0] == "far"} == {(i, j) for i in range(nb) for j in range(i + 1 + near(i), nb)}
    upd64 = {(t[1], t[2], t[3], t[4]) for t in tasks if t[0] == "upd64"}
    want64 = {(i, i, rh, ch) for i in range(1, nb) for rh, ch in ((0, 0), (0, 1), (1, 1))}
    want64 |= {(i, j, rh, ch) for i in range(1, nb) for j in range(i + 1, i + 1 + near(i)) for rh in (0, 1) for ch in (0, 1)}
    assert upd64 == want64
    # -- coverage: every block (i, j) takes the factor rows 0 .. i - 1 exactly once
    upd = {(t[1], t[2], t[3]) for t in tasks if t[0] == "UPD"}
    for i in range(nb):
        for j in range(i, nb):
            band_block = j - i <= band
            got = []
            for q in range(nb):
                if (q, i, j) in upd:
                    got += list(range(q * P, (q + 1) * P))
            if band_block and i >= 1:
                k_lo = next(t[5] for t in tasks if t[0] == "upd64" and t[1] == i and t[2] == j)
                got += list(range(k_lo, i))
            elif not band_block:
                got += list(range((i // P) * P, i))
            assert sorted(got) == list(range(i)), (i, j, got)
    # -- order: what a task waits for is in front of it
    def tr_pos(k, c64):                 # who makes the factor's block row k final at 64-column c64
        j, ch = divmod(c64, 2)
        if j - k <= band:
            return max(pos[("solve64", k, j, rh, ch)] for rh in (0, 1))
        return pos[next(t for t in tasks if t[0] == "far" and t[1] == k and t[2] == j)]
    diag_pos = lambda i: max(pos[t] for t in tasks if t[0] == "upd64" and t[1] == i and t[2] == i) if i else -1   # noqa: E731
    for t, k in pos.items():
        if t[0] == "UPD":
            q, bi, bj = t[1:]
            if q > 0:
                assert pos[("UPD", q - 1, bi, bj)] < k
            for kk in range(q * P, (q + 1) * P):
                for c in (2 * bi, 2 * bi + 1, 2 * bj, 2 * bj + 1):
                    assert tr_pos(kk, c) < k, (t, kk, c)
        elif t[0] == "upd64":
            i, j, rh, ch, k_lo, apt = t[1:]
            if apt > 0:
                assert pos[("UPD", apt - 1, i, j)] < k
            for kk in range(k_lo, i):
                assert tr_pos(kk, 2 * i + rh) < k and tr_pos(kk, 2 * j + ch) < k, (t, kk)
        elif t[0] == "solve64":
            i, j, rh, ch = t[1:]
            assert diag_pos(i) < k
            if i > 0:
                assert all(pos[next(u for u in tasks if u[0] == "upd64" and u[1:5] == (i, j, r2, ch))] < k for r2 in (0, 1))
        else:
            i, j, k_lo, apt = t[1:]
            assert diag_pos(i) < k
            if apt > 0:
                assert pos[("UPD", apt - 1, i, j)] < k
            for kk in range(k_lo, i):
                for c in (2 * i, 2 * i + 1, 2 * j, 2 * j + 1):
                    assert tr_pos(kk, c) < k, (t, kk, c)
