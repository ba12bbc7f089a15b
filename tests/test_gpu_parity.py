"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle and the committed golden
fixtures.  Tolerances are the ones stated in SURVEY.md 8(d):
    mu, jac : rtol 1e-10, atol 1e-12 * sigma_f * |alpha|_1
    var     : atol 1e-9 * sigma_f^2
    p1      : as mu ;  Q1 : rtol 1e-8
    ellipsoid algebra alone (identical mu/var/jac fed): rtol 1e-12
"""
import numpy as np
import pytest

from conftest import load_golden
from _helpers import hip_model, oracle_model, mu_atol, hyp_from, cached_oracle_model
from oracle import oracle_np as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(lib_built):
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"


# ------------------------------------------------------------------ MFMA tile in isolation
@pytest.mark.parametrize("M,N,K,mode", [(128, 128, 16, 0), (256, 384, 128, 0), (384, 384, 64, 1),
                                        (128, 512, 512, 2)])
def test_gemm_tn_matches_numpy(M, N, K, mode):
    import torch
    from safe_exploration_amd import _buffers as B
    from safe_exploration_amd._lib import lib, check
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M))          # asymmetric, transpose-detecting
    Bm = rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    if mode == 2:                            # B block-lower-triangular: B[k][n] = 0 for k < n0
        for n0 in range(0, N, 128):
            Bm[:n0, n0:n0 + 128] = 0.0
    dev = torch.device("cuda", 0)
    tA, tB, tC = B.as_dev(A, dev), B.as_dev(Bm, dev), B.as_dev(C0.copy(), dev)
    check(lib.sr_test_gemm_tn(0, B.ptr(tA), M, B.ptr(tB), N, B.ptr(tC), N, M, N, K, -0.5, 2.0, mode,
                              B.stream_ptr(dev)))
    got = B.to_numpy(tC)
    ref = -0.5 * A.T.dot(Bm) + 2.0 * C0
    if mode == 1:
        for m0 in range(0, M, 128):
            for n0 in range(0, N, 128):
                if n0 < m0:
                    ref[m0:m0 + 128, n0:n0 + 128] = C0[m0:m0 + 128, n0:n0 + 128]
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("seed,cond", [(0, 1.0), (1, 1e-6), (2, 1e3)])
def test_potrf_diag_block_matches_numpy(seed, cond):
    """The diagonal-block kernel of the model update (factor and inverse of one 128 x 128 block in one sweep): U against
    numpy's Cholesky, U^-1 and U^-T against the inverse of numpy's factor, the structural zeros of all three outputs
    (the blocks are read as full tiles by the GEMMs behind), and the first bad pivot of a matrix that is not positive
    definite.  The block sits inside a larger matrix (leading dimensions != 128, kb offset)."""
    import torch
    from safe_exploration_amd import _buffers as B
    from safe_exploration_amd._lib import lib, check
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((128, 160))
    A = M.dot(M.T) / 160 + cond * np.eye(128)
    dev = torch.device("cuda", 0)
    big = rng.standard_normal((128, 200))                   # lda = 200: the kernel must honour the leading dimension
    big[:, :128] = A
    tA = B.as_dev(big.copy(), dev)
    wt, w = B.empty((128, 136), dev).fill_(7.0), B.empty((128, 136), dev).fill_(7.0)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.sr_test_potrf_diag(0, B.ptr(tA), 200, B.ptr(wt), B.ptr(w), 136, B.ptr(info), 0, B.stream_ptr(dev)))
    assert int(info.item()) == 0
    got = B.to_numpy(tA)
    R = np.linalg.cholesky(A).T
    Ri = np.linalg.inv(R)
    scale = np.abs(Ri).max()
    np.testing.assert_allclose(got[:, :128], R, rtol=0, atol=2e-15 * np.abs(R).max() * max(1.0, 1.0 / np.sqrt(cond)))
    np.testing.assert_array_equal(got[:, 128:], big[:, 128:])                  # nothing outside the block touched
    gwt, gw = B.to_numpy(wt), B.to_numpy(w)
    np.testing.assert_allclose(gwt[:, :128], Ri, rtol=0, atol=1e-13 * scale * max(1.0, 1.0 / cond))
    np.testing.assert_array_equal(gw[:, :128], gwt[:, :128].T)
    assert np.all(np.tril(gwt[:, :128], -1) == 0) and np.all(np.triu(gw[:, :128], 1) == 0)
    assert np.all(gwt[:, 128:] == 7.0) and np.all(gw[:, 128:] == 7.0)
    # U^-1 U = I to rounding
    np.testing.assert_allclose(gwt[:, :128].dot(got[:, :128]), np.eye(128), rtol=0, atol=1e-12 * max(1.0, 1.0 / cond))
    # not positive definite from pivot 70 on: info = 70 (1-based), outputs stay finite (identity block)
    bad = A.copy()
    bad[69, 69] = -1.0
    tA = B.as_dev(bad, dev)
    info.zero_()
    wt2, w2 = B.empty((128, 128), dev), B.empty((128, 128), dev)
    check(lib.sr_test_potrf_diag(0, B.ptr(tA), 128, B.ptr(wt2), B.ptr(w2), 128, B.ptr(info), 0, B.stream_ptr(dev)))
    assert int(info.item()) == 70
    assert np.array_equal(B.to_numpy(wt2), np.eye(128)) and np.array_equal(B.to_numpy(w2), np.eye(128))
    # the first bad pivot wherever it sits in a 16-pivot tile (first / last pivot of a tile, first / last tile), also when
    # it is a NaN, and when a second bad pivot follows (the branch-free pivot chain finds it from the NaNs it leaves behind)
    for pos, val, extra in ((0, -1.0, None), (15, 0.0, None), (16, -2.0, None), (31, np.nan, None), (112, -1.0, 120),
                            (127, -1e-300, None), (5, -1.0, 6)):
        bad = A.copy()
        bad[pos, pos] = val
        if extra is not None:
            bad[extra, extra] = -1.0
        tA = B.as_dev(bad, dev)
        info.zero_()
        check(lib.sr_test_potrf_diag(0, B.ptr(tA), 128, B.ptr(wt2), B.ptr(w2), 128, B.ptr(info), 0, B.stream_ptr(dev)))
        assert int(info.item()) == pos + 1, (pos, val, int(info.item()))
        assert np.array_equal(B.to_numpy(wt2), np.eye(128)) and np.array_equal(B.to_numpy(w2), np.eye(128))
        assert np.array_equal(B.to_numpy(tA), np.eye(128))


@pytest.mark.parametrize("N,n_s,panel", [(300, 2, 0), (900, 2, 0), (1500, 4, 0), (2100, 2, 2), (2100, 2, 4), (3300, 1, 0)])
def test_pipelined_model_update_equals_the_plain_chain(N, n_s, panel):
    """Round 6: the block step of the Cholesky cut at its dependencies and dealt to three streams that hand over through
    device counters (sr_capi_update.hip, `if (pipe)`) runs the SAME tiles in the same k order as the one chain of launches:
    alpha and U^-1 must come out bit for bit, at panel boundaries that fall everywhere (sizes with 3 .. 26 blocks, panels
    of 2, 3, 4), repeatedly on one handle (the counters carry epochs), and the library must say that it did run pipelined."""
    import torch
    from safe_exploration_amd import workload, SimpleGPModel
    prob = workload.make_problem(40 + N, N, n_s, 1, 4)
    out = {}
    for pipe in (1, 0, 2, 1):
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
        gp.set_fact_panel(panel)
        gp.set_fact_pipeline(pipe)
        for rep in range(3 if pipe else 1):
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            assert gp.fact_pipelined() == bool(pipe) or N <= 256, (N, pipe)      # (fewer than 3 blocks: nothing to pipeline)
            alpha, wt = gp.export_state()
            got = (alpha.cpu().numpy().copy(), wt.cpu().numpy().copy())
            if "ref" in out:
                assert np.array_equal(got[0], out["ref"][0]), (N, pipe, rep)
                assert np.array_equal(got[1], out["ref"][1]), (N, pipe, rep)
            else:
                out["ref"] = got
        del gp
    torch.cuda.synchronize()
    # and the posterior identity at the training inputs (the chain did factor THIS matrix)
    gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    s2n = prob["noise_var"] + 1e-5 + 1e-8
    mu, _ = gp.predict(prob["Z"][:256])
    assert np.abs(mu + s2n[None, :] * gp.beta[:256] - prob["Y"][:256]).max() < 1e-9


@pytest.mark.parametrize("N,n_s,panel", [(300, 2, 0), (500, 2, 0), (900, 2, 2), (1500, 4, 0), (2100, 2, 2), (2100, 3, 5), (2500, 2, 64),
                                         (3300, 1, 0)])
def test_tile_flow_model_update_agrees_with_the_chain_of_launches(N, n_s, panel):
    """Round 6: the whole Cholesky as ONE resident kernel of tile tasks plus a resident diagonal-block workgroup per output,
    dependencies through device counters (csrc/sr_flow.hip; opt-in: set_fact_pipeline(3)).  It sums in another order than the
    chain of launches (left-looking band, panels of its own), so alpha and U^-1 agree to rounding, not bit for bit -- but every
    tile has ONE order of summation whatever the workgroups' timing: repeated updates on one handle (the counters carry epochs)
    and a second handle must reproduce the first bit for bit, which a stale read of a tile another XCD rewrote would not.
    Panels of 2 / 3 / 5 blocks and none (64 > the number of block rows: everything left-looking), 1 - 4 outputs, 3 .. 26 blocks; below 3 blocks the
    knob is accepted and the chain of launches runs."""
    import torch
    from safe_exploration_amd import workload, SimpleGPModel
    prob = workload.make_problem(70 + N, N, n_s, 1, 4)
    ref = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
    ref.set_fact_pipeline(-1)
    ref.train(prob["Z"], prob["Y"], opt_hyp=False)
    assert ref.fact_route() == 0
    a0, w0 = ref.export_state()
    a0, w0 = a0.cpu().numpy().copy(), w0.cpu().numpy().copy()
    del ref
    first = None
    for handle in range(2):
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
        gp.set_fact_panel(panel)
        gp.set_fact_pipeline(3)
        for rep in range(3 if handle == 0 else 1):
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            assert gp.fact_route() == (4 if N > 256 else 0), (N, gp.fact_route())
            alpha, wt = gp.export_state()
            got = (alpha.cpu().numpy().copy(), wt.cpu().numpy().copy())
            if first is None:
                first = got
                assert np.abs(got[0] - a0).max() <= 1e-9 * np.abs(a0).max()
                assert np.abs(got[1] - w0).max() <= 1e-10 * np.abs(w0).max()
            else:
                assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]), (N, handle, rep)
        if handle == 0:
            # the posterior identity at the training inputs (the flow did factor THIS matrix)
            s2n = prob["noise_var"] + 1e-5 + 1e-8
            mu, _ = gp.predict(prob["Z"][:256])
            assert np.abs(mu + s2n[None, :] * gp.beta[:256] - prob["Y"][:256]).max() < 1e-9
            if N > 256:
                # its diagnostics: every task of the plan was counted once
                import ctypes
                from safe_exploration_amd import _lib
                nb = gp._handle.Np // 128
                buf = (ctypes.c_uint * (24 + n_s * nb))()
                assert _lib.lib.sr_gp_flow_stats(gp._handle.h, buf, len(buf)) == len(buf)
                sg, tot = (ctypes.c_int * (4 * (nb + 1)))(), (ctypes.c_long * 4)()
                p_eff = panel if panel > 0 else (2 if nb <= 12 else (3 if nb <= 28 else (4 if nb <= 36 else (6 if nb <= 44 else 8))))
                assert _lib.lib.sr_test_flow_plan(nb, 2, p_eff, sg, tot) == 0
                assert sum(buf[4 * k] for k in range(6)) == tot[3] * n_s
                ticks = [buf[24 + k] for k in range(nb)]
                assert all(b > a for a, b in zip(ticks, ticks[1:]))
        del gp
    torch.cuda.synchronize()


def test_tile_flow_that_fails_on_the_device_is_repeated_by_launches():
    """Every wait of the tile flow has a time-out (0.25 s), and a wait that runs into it raises a status word that makes
    everybody leave: the resident kernels END, whatever happened.  The update is then repeated by launches -- the caller gets
    the right model, late -- and the tile flow rests for the process' next 16 updates (32, 64, .. after further failures;
    sr_test_flow_fail: the diagonal-block workgroups are given an epoch nobody publishes)."""
    import time
    from safe_exploration_amd import workload, SimpleGPModel, _lib
    N, n_s = 900, 2
    prob = workload.make_problem(11, N, n_s, 1, 4)
    s2n = prob["noise_var"] + 1e-5 + 1e-8
    gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
    gp.set_fact_pipeline(3)
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    assert gp.fact_route() == 4
    try:
        assert _lib.lib.sr_test_flow_fail(1) == 0
        t0 = time.perf_counter()
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        took = time.perf_counter() - t0
        assert gp.fact_route() == 0 and took < 5.0, (gp.fact_route(), took)
        mu, _ = gp.predict(prob["Z"][:128])
        assert np.abs(mu + s2n[None, :] * gp.beta[:128] - prob["Y"][:128]).max() < 1e-9
        # once failed: the flow rests for 16 would-be flows of the process (the repeat was the first), then it is back
        rested = 0
        for _ in range(40):
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            if gp.fact_route() == 4:
                break
            rested += 1
        assert rested == 15 and gp.fact_route() == 4, (rested, gp.fact_route())
    finally:
        assert _lib.lib.sr_test_flow_fail(0) == 0
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    assert gp.fact_route() == 4
    mu, _ = gp.predict(prob["Z"][:128])
    assert np.abs(mu + s2n[None, :] * gp.beta[:128] - prob["Y"][:128]).max() < 1e-9


def test_tile_flow_reports_a_matrix_that_is_not_positive_definite():
    """A breakdown inside the resident diagonal-block workgroup (a NaN among the training inputs: the pivot of its row is not
    positive) is reported like the launched kernel's -- LinAlgError naming the same pivot --, nothing hangs, and the handle
    takes a good model afterwards, by the tile flow again."""
    from safe_exploration_amd import workload, SimpleGPModel
    N, n_s = 900, 2
    prob = workload.make_problem(7, N, n_s, 1, 4)
    Z = prob["Z"].copy()
    Z[300, 0] = np.nan
    res = {}
    for knob in (-1, 3):
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
        gp.set_fact_pipeline(knob)
        with pytest.raises(np.linalg.LinAlgError) as exc:
            gp.train(Z, prob["Y"], opt_hyp=False)
        res[knob] = str(exc.value).split("breakdown:")[-1]
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        assert gp.fact_route() == (4 if knob == 3 else 0)
        mu, _ = gp.predict(prob["Z"][:64])
        s2n = prob["noise_var"] + 1e-5 + 1e-8
        assert np.abs(mu + s2n[None, :] * gp.beta[:64] - prob["Y"][:64]).max() < 1e-9
        del gp
    assert res[-1] == res[3], res


# ------------------------------------------------------------------ GP fit + predict
@pytest.mark.parametrize("name,n_s,n_u", [("gp_pend.npz", 2, 1), ("gp_cart.npz", 4, 1)])
def test_predict_matches_golden_and_oracle(name, n_s, n_u):
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    om = oracle_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])
    at = mu_atol(om)
    sf2 = float(np.max(g["signal_var"]))
    # cached posterior state
    np.testing.assert_allclose(gp.beta, g["beta"], rtol=1e-8, atol=1e-9 * np.abs(g["beta"]).max())
    for d in range(n_s):
        np.testing.assert_allclose(gp.inv_K[d], om["inv_K"][d], rtol=1e-7,
                                   atol=1e-9 * np.abs(om["inv_K"][d]).max())
    mu, var, jac = gp.predict(g["x_new"], None, True)
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-10, atol=max(at, 1e-12))
    np.testing.assert_allclose(jac, g["jac"], rtol=1e-10, atol=max(10 * at, 1e-11))
    np.testing.assert_allclose(var, g["var"], rtol=0, atol=1e-9 * sf2)
    # the fixture values produced by the reference's own gp_pred formulas
    np.testing.assert_allclose(mu, g["ref_mu"], rtol=1e-10, atol=max(at, 1e-12))
    np.testing.assert_allclose(var, g["ref_var"], rtol=0, atol=1e-9 * sf2)
    # both call shapes + the single-query 3-tuple
    mu2, var2 = gp.predict(g["x_new"][:, :n_s], g["x_new"][:, n_s:])
    np.testing.assert_array_equal(mu2, mu)
    np.testing.assert_array_equal(var2, var)
    m1, s1, j1 = gp(g["x_new"][3:4, :n_s], g["x_new"][3:4, n_s:])
    assert m1.shape == (n_s, 1) and s1.shape == (n_s, 1) and j1.shape == (n_s, n_s + n_u)
    np.testing.assert_allclose(m1[:, 0], g["mu"][3], rtol=1e-10, atol=max(at, 1e-12))
    np.testing.assert_allclose(j1, g["jac"][3], rtol=1e-10, atol=max(10 * at, 1e-11))
    with pytest.raises(NotImplementedError):
        gp(g["x_new"][:2, :n_s], g["x_new"][:2, n_s:])


@pytest.mark.parametrize("N,n_s,n_u,T", [(1, 2, 1, 5), (2, 2, 1, 3), (127, 2, 1, 130), (129, 4, 1, 257),
                                         (300, 3, 2, 1), (700, 2, 1, 1000)])
def test_predict_ragged_sizes(N, n_s, n_u, T):
    """padding edges: N around the 128 block, T around the 128/256 tiles, T = 1, N = 1."""
    syn = orc.make_synthetic(7 * N + T, N, n_s, n_u, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu, var, jac = gp.predict(x, None, True)
    rmu, rvar, rjac = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"],
                                     om["signal_var"], True)
    at = max(mu_atol(om), 1e-12)
    np.testing.assert_allclose(mu, rmu, rtol=1e-9, atol=at)
    np.testing.assert_allclose(jac, rjac, rtol=1e-9, atol=10 * at)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    # second algebraic route of the oracle agrees too
    _, cvar = orc.gp_predict_chol(x, om["Z"], om["beta"], om["chol"], om["lengthscale"], om["signal_var"])
    np.testing.assert_allclose(var, cvar, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n_out,N", [(9, 150), (11, 300), (17, 140)])
def test_more_outputs_than_factorisation_slots(n_out, N):
    """n_out > 8: the outputs share the 8 factorisation slots (streams + scratch) round-robin; every output's
    posterior still equals the oracle's."""
    rng = np.random.default_rng(n_out)
    D, T = 3, 40
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, n_out))
    ls = rng.uniform(0.5, 1.5, (n_out, D))
    sf2 = rng.uniform(0.5, 1.5, n_out)
    noise = np.full(n_out, 1e-2 + 1e-5)
    from safe_exploration_amd import SimpleGPModel
    gp = SimpleGPModel(n_out, 2, 1, kern_types=["rbf"] * n_out, hyp=hyp_from(ls, sf2, noise))
    gp.train(Z, Y, opt_hyp=False)
    om = oracle_model(Z, Y, ls, sf2, noise)
    x = rng.uniform(-0.8, 0.8, (T, D))
    mu, var, jac = gp.predict(x, None, True)
    rmu, rvar, rjac = orc.gp_predict(x, Z, om["beta"], om["inv_K"], ls, sf2)
    at = max(mu_atol(om), 1e-12)
    np.testing.assert_allclose(gp.beta, om["beta"], rtol=1e-7, atol=1e-9 * np.abs(om["beta"]).max())
    np.testing.assert_allclose(mu, rmu, rtol=1e-10, atol=at)
    np.testing.assert_allclose(jac, rjac, rtol=1e-10, atol=10 * at)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9 * sf2.max())


def test_predict_empty_batch():
    syn = orc.make_synthetic(5, 40, 2, 1, 4)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    mu, var = gp.predict(np.empty((0, 3)))
    assert mu.shape == (0, 2) and var.shape == (0, 2)


def test_interpolation_property():
    """mu(z_i) -> y_i and var(z_i) -> O(noise) as noise -> 0."""
    rng = np.random.default_rng(3)
    Z = rng.uniform(-1, 1, (50, 3))
    Y = np.sin(Z.dot(rng.standard_normal((3, 2))))
    ls = np.full((2, 3), 0.7)
    gp = hip_model(Z, Y, ls, np.ones(2), np.full(2, 1e-6 + 1e-5), 2, 1)
    mu, var = gp.predict(Z)
    assert np.abs(mu - Y).max() < 1e-3
    assert var.max() < 1e-4 and var.min() > 0


def test_not_positive_definite_is_reported():
    Z = np.zeros((4, 3))                       # identical points, ~zero noise -> singular K
    from safe_exploration_amd import SimpleGPModel
    gp = SimpleGPModel(2, 2, 1, hyp=[{"lengthscale": np.ones(3), "variance": 1.0,
                                      "noise_variance": -1e-5 - 1e-8 - 1e-3}] * 2)
    with pytest.raises(np.linalg.LinAlgError):
        gp.train(Z, np.zeros((4, 2)), opt_hyp=False)


# ------------------------------------------------------------------ ellipsoid kernel in isolation
@pytest.mark.parametrize("name", ["reach_pend.npz", "reach_cart.npz", "reach_n3u2.npz"])
def test_ellipsoid_step_vs_reference_golden(name):
    """identical (mu,var,jac) fed -> pure small-matrix algebra, rtol 1e-12 vs the REFERENCE's output."""
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden(name)
    for tag, a, b in (("id", None, None), ("lin", g["a_lin"], g["b_lin"])):
        p1, q1 = reach.ellipsoid_step_batch(g["p"], g["k_ff"], g["mu"], g["var"], g["jac"], g["l_mu"],
                                            g["l_sigma"], None, None, float(g["c_safety"]), a, b,
                                            check_bounds=True)
        np.testing.assert_allclose(p1, g["p1_point_" + tag], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(q1, g["q1_point_" + tag], rtol=1e-12, atol=1e-16)
        p1, q1 = reach.ellipsoid_step_batch(g["p"], g["k_ff"], g["mu"], g["var"], g["jac"], g["l_mu"],
                                            g["l_sigma"], g["Q"], g["k_fb"], float(g["c_safety"]), a, b,
                                            check_bounds=True)
        np.testing.assert_allclose(p1, g["p1_ell_" + tag], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(q1, g["q1_ell_" + tag], rtol=1e-12, atol=1e-15)


def test_anchor_known_answers():
    """SURVEY 8c worked anchor (values produced by the imported reference functions)."""
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden("anchor.npz")
    const = lambda s, a: (g["mu"], g["var"], g["jac"])      # a non-HIP StateSpaceModel stand-in
    l = g["l"]
    pp, qp = reach.onestep_reachability(g["p"], const, g["k_ff"], l, l, None, None, 2.0, 0)
    np.testing.assert_allclose(pp, g["p_point"], rtol=1e-13)
    np.testing.assert_allclose(qp, np.diag([0.08, 0.32]), rtol=1e-13)      # hand check n_s (c sigma)^2
    pe, qe = reach.onestep_reachability(g["p"], const, g["k_ff"], l, l, g["Q"], g["k_fb"], 2.0, 0)
    np.testing.assert_allclose(pe, g["p_ell"], rtol=1e-13)
    np.testing.assert_allclose(qe, g["q_ell"], rtol=1e-12)
    np.testing.assert_allclose(qe, [[0.605462201964151, 0.158620529130949],
                                    [0.158620529130949, 0.943186225924461]], rtol=1e-12)
    pl, ql = reach.onestep_reachability(g["p"], const, g["k_ff"], l, l, g["Q"], g["k_fb"], 2.0, 0,
                                        g["a2"], g["b2"])
    np.testing.assert_allclose(pl, g["p_lin"], rtol=1e-13)
    np.testing.assert_allclose(ql, g["q_lin"], rtol=1e-12)
    h = np.vstack((np.eye(2), -np.eye(2)))
    d = reach.lin_ellipsoid_safety_distance(pe, qe, h, np.ones((4, 1)), 1.0)
    np.testing.assert_allclose(d, g["d_safety"], rtol=1e-12)
    # demo known answer of the reference: p=0, Q=I/4, H=[e1;-e1], h=.5 -> d = 0
    d0 = reach.lin_ellipsoid_safety_distance(np.zeros((3, 1)), 0.25 * np.eye(3),
                                             np.array([[1., 0, 0], [-1., 0, 0]]), 0.5 * np.ones((2, 1)))
    np.testing.assert_allclose(d0, 0.0, atol=1e-15)


def test_remainder_golden():
    from safe_exploration_amd import utils
    g = load_golden("remainder.npz")
    for tag in "1234":
        um, us = utils.compute_remainder_overapproximations(g["q_" + tag], g["k_fb_" + tag],
                                                            g["l_mu_" + tag], g["l_sigma_" + tag])
        np.testing.assert_allclose(um, g["u_mu_" + tag], rtol=1e-12)
        np.testing.assert_allclose(us, g["u_sigma_" + tag], rtol=1e-12)
        um, us = utils.compute_remainder_overapproximations(g["q_" + tag], 0 * g["k_fb_" + tag],
                                                            g["l_mu_" + tag], g["l_sigma_" + tag])
        np.testing.assert_allclose(um, g["u_mu_k0_" + tag], rtol=1e-12)
        # k_fb = 0 => r^2 = lambda_max(Q)
        np.testing.assert_allclose(um, g["l_mu_" + tag] * np.linalg.eigvalsh(g["q_" + tag])[-1], rtol=1e-12)


def test_bad_box_raises_like_reference():
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden("anchor.npz")
    const = lambda s, a: (g["mu"], g["var"], g["jac"])
    with pytest.raises(AssertionError):       # l_mu = 0 -> zero-width box (utils_ellipsoid.py:227)
        reach.onestep_reachability(g["p"], const, g["k_ff"], 0 * g["l"], g["l"], g["Q"], g["k_fb"], 2.0, 0)


def test_bad_box_is_counted_on_every_staging_route():
    """A zero-width box must raise for NumPy callers whether the batch is copied to the device or -- a handful of numbers --
    read and written in the pinned staging blocks by the kernels themselves (the violation counter then lives in pinned host
    memory and is incremented from the device), and for device tensors."""
    import torch
    from safe_exploration_amd import gp_reachability as reach
    syn = orc.make_synthetic(5, 120, 2, 1, 300)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    l, l0 = np.array([0.05, 0.02]), np.zeros(2)
    for T in (1, 3, 300):                                    # 1, 3: zero-copy; 300: one pinned block copied
        args = (syn["p"][:T], gp, syn["k_ff"][:T])
        reach.onestep_reachability_batch(*args, l, l, syn["Q"][:T], syn["k_fb"][:T], 2.0, check_bounds=True)
        with pytest.raises(AssertionError):
            reach.onestep_reachability_batch(*args, l0, l, syn["Q"][:T], syn["k_fb"][:T], 2.0, check_bounds=True)
    dev = lambda a: torch.as_tensor(a, device="cuda:0")
    with pytest.raises(AssertionError):
        reach.onestep_reachability_batch(dev(syn["p"][:2]), gp, dev(syn["k_ff"][:2]), l0, l, dev(syn["Q"][:2]), dev(syn["k_fb"][:2]),
                                         2.0, check_bounds=True)


# ------------------------------------------------------------------ fused GP + ellipsoid
@pytest.mark.parametrize("name,n_s,n_u", [("reach_pend.npz", 2, 1), ("reach_cart.npz", 4, 1),
                                          ("reach_n3u2.npz", 3, 2)])
def test_onestep_and_multistep_vs_reference_golden(name, n_s, n_u):
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    c = float(g["c_safety"])
    for tag, a, b in (("id", None, None), ("lin", g["a_lin"], g["b_lin"])):
        p1, q1 = reach.onestep_reachability_batch(g["p"], gp, g["k_ff"], g["l_mu"], g["l_sigma"], None,
                                                  None, c, a, b, check_bounds=True)
        np.testing.assert_allclose(p1, g["p1_point_" + tag], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(q1, g["q1_point_" + tag], rtol=1e-8, atol=1e-14)
        p1, q1, var = reach.onestep_reachability_batch(g["p"], gp, g["k_ff"], g["l_mu"], g["l_sigma"],
                                                       g["Q"], g["k_fb"], c, a, b, check_bounds=True,
                                                       return_var=True)
        np.testing.assert_allclose(p1, g["p1_ell_" + tag], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(q1, g["q1_ell_" + tag], rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(var, g["var"], rtol=0, atol=1e-9 * float(np.max(g["signal_var"])))
    # single-query API with the reference's shapes
    p_1, q_1 = reach.onestep_reachability(g["p"][0][:, None], gp, g["k_ff"][0][:, None], g["l_mu"],
                                          g["l_sigma"], g["Q"][0], g["k_fb"][0], c, 0)
    assert p_1.shape == (n_s, 1) and q_1.shape == (n_s, n_s)
    np.testing.assert_allclose(q_1, g["q1_ell_id"][0], rtol=1e-8)
    # multi-step chains (errors compound over H steps)
    pa, qa = reach.multistep_reachability_batch(g["ms_p0"], gp, g["ms_k_fb"], g["ms_k_ff"], g["l_mu"],
                                                g["l_sigma"], None, c, g["a_lin"], g["b_lin"], None)
    np.testing.assert_allclose(pa, g["ms_p_all"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(qa, g["ms_q_all"], rtol=1e-6, atol=1e-12)
    Tm = g["ms_p0"].shape[0]
    pa, qa = reach.multistep_reachability_batch(g["ms_p0"], gp, g["ms_k_fb"], g["ms_k_ff"], g["l_mu"],
                                                g["l_sigma"], g["Q"][:Tm], c, g["a_lin"], g["b_lin"],
                                                g["k_fb"][:Tm])
    np.testing.assert_allclose(pa, g["ms_p_all_q0"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(qa, g["ms_q_all_q0"], rtol=1e-6, atol=1e-12)
    # single-trajectory API: (p_new, q_new, p_all, q_all) like gp_reachability.py:212
    pn, qn, p_all, q_all = reach.multistep_reachability(g["ms_p0"][0][:, None], gp, g["ms_k_fb"][0],
                                                        g["ms_k_ff"][0], g["l_mu"], g["l_sigma"], None,
                                                        c, 0, g["a_lin"], g["b_lin"], None)
    np.testing.assert_allclose(p_all, g["ms_p_all"][0], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(qn, g["ms_q_all"][0][-1], rtol=1e-6)
    assert pn.shape == (n_s, 1)
    # safety distance on the results
    d = reach.lin_ellipsoid_safety_distance_batch(g["p1_ell_id"], g["q1_ell_id"], g["h_mat"], g["h_vec"], c)
    np.testing.assert_allclose(d, g["d_safety"], rtol=1e-12, atol=1e-14)


def test_release_scratch_then_refit():
    """sr_gp_release_scratch frees what update / append keep; the next update allocates again and gives the same model."""
    syn = orc.make_synthetic(5, 700, 2, 1, 64)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu0, var0 = gp.predict(x)
    beta0 = gp.beta.copy()
    gp.release_scratch()
    mu1, var1 = gp.predict(x)                       # the model itself is untouched
    np.testing.assert_array_equal(mu0, mu1)
    np.testing.assert_array_equal(var0, var1)
    gp.train(syn["Z"], syn["Y"], opt_hyp=False)
    np.testing.assert_array_equal(gp.beta, beta0)
    gp.update_model(syn["Z"][:20] + 0.01, syn["Y"][:20], opt_hyp=False, replace_old=False)
    gp.release_scratch()
    assert gp.predict(x)[0].shape == mu0.shape


def test_chunking_is_invisible():
    """results do not depend on the internal chunk size (ragged last chunk included)."""
    from safe_exploration_amd import gp_reachability as reach
    syn = orc.make_synthetic(77, 200, 2, 1, 1000)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    l = np.array([0.05, 0.02])
    ref = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0)
    gp.set_chunk(384)
    got = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0)
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-12, atol=1e-16)


def test_full_size_properties():
    """BASELINE configs[1] size (N=2000, T=65536): size-independent properties.
    (i) a 4096-query sample agrees with the oracle, (ii) permuting the queries permutes the outputs
    bit-exactly, (iii) Q1 symmetric positive definite, (iv) Q1 is monotone in c_safety."""
    from safe_exploration_amd import gp_reachability as reach
    N, T = 2000, 65536
    syn = orc.make_synthetic(2, N, 2, 1, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    l = np.array([0.05, 0.02])
    p1, q1, var = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"],
                                                   2.0, return_var=True)
    assert np.all(np.isfinite(q1))
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    idx = np.random.default_rng(0).choice(T, 4096, replace=False)
    rp, rq, rvar = orc.onestep_reachability_vectorised(om, syn["p"][idx], syn["Q"][idx], syn["k_ff"][idx],
                                                       syn["k_fb"][idx], l, l, 2.0, np.eye(2), np.zeros((2, 1)))
    np.testing.assert_allclose(p1[idx], rp, rtol=1e-9, atol=max(mu_atol(om), 1e-12))
    np.testing.assert_allclose(var[idx], rvar, rtol=0, atol=1e-9)
    np.testing.assert_allclose(q1[idx], rq, rtol=1e-8, atol=1e-12)
    perm = np.random.default_rng(1).permutation(T)
    p1p, q1p = reach.onestep_reachability_batch(syn["p"][perm], gp, syn["k_ff"][perm], l, l, syn["Q"][perm],
                                                syn["k_fb"][perm], 2.0)
    np.testing.assert_array_equal(p1p, p1[perm])
    np.testing.assert_array_equal(q1p, q1[perm])
    np.testing.assert_allclose(q1, np.swapaxes(q1, 1, 2), rtol=1e-12, atol=1e-15)
    assert np.linalg.eigvalsh(q1).min() > 0
    _, q1c = reach.onestep_reachability_batch(syn["p"][:2048], gp, syn["k_ff"][:2048], l, l, syn["Q"][:2048],
                                              syn["k_fb"][:2048], 3.0)
    assert np.all(np.trace(q1c, axis1=1, axis2=2) > np.trace(q1[:2048], axis1=1, axis2=2))


def test_var_kernel_variants_agree_bitwise():
    """The pipelined main loop of round 5 (variant 3) is the arithmetic of the loop of rounds 1 - 4 (variant 1) in the same
    order: equal bit for bit.  Variant 4 (the product's) leaves out the structural zeros of the diagonal blocks and deals
    the rows of a block to the wavefronts differently: same numbers, another order of summation.  Variant 5 (round 6)
    runs the tiles of 4 two per workgroup: bit for bit 4.
    The A/B forms live in the LAB build (scripts/_bin/libsafereach_lab.so, `make lab`): the test runs itself once more in
    a process that loads that library; the product answers SR_EUNSUPPORTED for them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lab = os.path.join(root, "scripts", "_bin", "libsafereach_lab.so")
    if os.environ.get("SAFEREACH_LIB") != lab:
        probe = hip_model(*[orc.make_synthetic(5, 40, 2, 1, 4)[k] for k in ("Z", "Y", "lengthscale", "signal_var", "noise_var")], 2, 1)
        with pytest.raises(NotImplementedError):
            probe.set_var_variant(1)
        probe.set_var_variant(4)
        if not os.path.exists(lab):
            pytest.skip("lab build missing (make -C safe_exploration_amd/csrc lab)")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                            "test_var_kernel_variants_agree_bitwise"], env=dict(os.environ, SAFEREACH_LIB=lab), cwd=root,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
        return
    syn = orc.make_synthetic(91, 700, 2, 1, 3000)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp.set_var_variant(1)
    mu0, var0 = gp.predict(x)
    gp.set_var_variant(3)
    mu1, var1 = gp.predict(x)
    np.testing.assert_array_equal(var0, var1)
    np.testing.assert_array_equal(mu0, mu1)
    gp.set_var_variant(4)
    mu2, var2 = gp.predict(x)
    np.testing.assert_array_equal(mu0, mu2)
    np.testing.assert_allclose(var2, var0, rtol=0, atol=1e-13)
    gp.set_var_variant(5)                                  # (round 6) pairs of row blocks per workgroup: the tiles of variant 4
    _, var5 = gp.predict(x)
    np.testing.assert_array_equal(var5, var2)
    # every count of k-tiles modulo the pair structure of the pipelined loop, with and without a diagonal-block walk
    # (front padding 0 .. 127 rows moves the first k-tile; row block 0 of a padded model has fewer than eight k-tiles)
    for N in (1, 17, 33, 100, 128, 129, 145, 161, 250, 257, 300, 383, 400):
        s2 = orc.make_synthetic(200 + N, N, 2, 1, 150)
        g2 = hip_model(s2["Z"], s2["Y"], s2["lengthscale"], s2["signal_var"], s2["noise_var"], 2, 1)
        g2.set_small_path(0)                               # the plain tiles, whatever the size
        x2 = np.hstack((s2["p"], s2["k_ff"]))
        g2.set_var_variant(1)
        _, v1 = g2.predict(x2)
        g2.set_var_variant(3)
        _, v3 = g2.predict(x2)
        g2.set_var_variant(4)
        _, v4 = g2.predict(x2)
        g2.set_var_variant(5)
        _, v5 = g2.predict(x2)
        np.testing.assert_array_equal(v4, v5, err_msg="N=%d" % N)
        np.testing.assert_array_equal(v1, v3, err_msg="N=%d" % N)
        np.testing.assert_allclose(v4, v3, rtol=0, atol=1e-13, err_msg="N=%d" % N)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    _, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(var2, rvar, rtol=0, atol=1e-9)


@pytest.mark.parametrize("name,n_s,n_u", [("gp_pend.npz", 2, 1), ("gp_cart.npz", 4, 1)])
def test_linearize_predict_second_order(name, n_s, n_u):
    """A10: d var/dx and Hessian of mu for a single query (the CasADi Jacobian callback's inputs)."""
    from safe_exploration_amd import utils
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    om = oracle_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])
    x0 = g["x_new"][0]
    mu, var, jm, jv, hm = gp.linearize_predict(x0[None, :n_s], x0[None, n_s:], True, False)
    D = n_s + n_u
    assert mu.shape == (n_s, 1) and var.shape == (n_s, 1) and jm.shape == (n_s, D)
    assert jv.shape == (n_s, D) and hm.shape == (n_s, D, D)
    np.testing.assert_allclose(mu[:, 0], g["mu"][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(jm, g["jac"][0], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(jv, g["jac_var0"], rtol=1e-7, atol=1e-9 * float(np.max(g["signal_var"])))
    np.testing.assert_allclose(hm, g["hess_mu0"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(hm, np.swapaxes(hm, 1, 2), rtol=0, atol=0)
    # stacked layout the evaluator builds (state_space_models.py:407-410)
    stacked = np.vstack((jm, jv, utils.reshape_derivatives_3d_to_2d(hm)))
    assert stacked.shape == (2 * n_s + n_s * D, D)
    # another query + reverse mode == seed^T J
    x1 = g["x_new"][5]
    rjv, rhm = orc.gp_linearize_extras(x1, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    out = gp.linearize_predict(x1[None, :n_s], x1[None, n_s:], True)
    np.testing.assert_allclose(out[3], rjv, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(out[4], rhm, rtol=1e-9, atol=1e-10)
    seed = np.random.default_rng(0).standard_normal(2 * n_s + n_s * D)
    gs, ga = gp.get_linearize_reverse(seed)
    ref = seed.dot(np.vstack((out[2], out[3], utils.reshape_derivatives_3d_to_2d(out[4]))))
    np.testing.assert_allclose(np.vstack((gs, ga))[:, 0], ref, rtol=1e-12)
    # base-class predict(states, actions, jacobians=True): 4 outputs incl. d var/dx, batched by looping
    m4, v4, jm4, jv4 = gp.predict(g["x_new"][:3, :n_s], g["x_new"][:3, n_s:], True)
    assert jv4.shape == (3, n_s, D)
    np.testing.assert_allclose(jv4[0], g["jac_var0"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(m4, g["mu"][:3], rtol=1e-10, atol=1e-12)
    with pytest.raises(NotImplementedError):
        gp.linearize_predict(g["x_new"][:2, :n_s], g["x_new"][:2, n_s:], True)


def test_config1_reference_sizes():
    """BASELINE configs[0]: pendulum, N=200 training points, batch predict at 1024 query states."""
    syn = orc.make_synthetic(1, 200, 2, 1, 1024)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu, var, jac = gp.predict(x, None, True)
    rmu, rvar, rjac = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    at = max(mu_atol(om), 1e-12)
    np.testing.assert_allclose(mu, rmu, rtol=1e-10, atol=at)
    np.testing.assert_allclose(jac, rjac, rtol=1e-10, atol=10 * at)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    from safe_exploration_amd import gp_reachability as reach
    l = np.array([0.05, 0.02])
    p1, q1 = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0)
    rp, rq, _ = orc.onestep_reachability_vectorised(om, syn["p"], syn["Q"], syn["k_ff"], syn["k_fb"], l, l, 2.0,
                                                    np.eye(2), np.zeros((2, 1)))
    np.testing.assert_allclose(p1, rp, rtol=1e-10, atol=at)
    np.testing.assert_allclose(q1, rq, rtol=1e-8, atol=1e-14)


def test_many_chunks_million_queries():
    """config 5 per-GPU share: 1,048,576 queries stream through 16 chunks of the bounded workspace;
    spot-check against the oracle and check chunk-boundary rows against a direct small call."""
    from safe_exploration_amd import gp_reachability as reach
    import torch
    N, T = 300, 1 << 20
    syn = orc.make_synthetic(55, N, 2, 1, 4096)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    reps = T // 4096
    dev = gp.device
    tile = lambda a: torch.from_numpy(a).to(dev).repeat((reps,) + (1,) * (a.ndim - 1))
    l = np.array([0.05, 0.02])
    p1, q1 = reach.onestep_reachability_batch(tile(syn["p"]), gp, tile(syn["k_ff"]), l, l, tile(syn["Q"]),
                                              tile(syn["k_fb"]), 2.0)
    assert p1.shape == (T, 2) and q1.shape == (T, 2, 2)
    ref_p, ref_q = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0)
    q1 = q1.cpu().numpy().reshape(reps, 4096, 2, 2)
    # (different N-split of the mu/J partial sums between the two batch sizes: last-bit differences)
    np.testing.assert_allclose(q1[0], ref_q, rtol=1e-10, atol=1e-13)
    np.testing.assert_array_equal(q1[reps - 1], q1[0])         # every replica of the block is identical
    np.testing.assert_array_equal(q1[reps // 2], q1[0])


@pytest.mark.parametrize("N,T", [(100, 1), (300, 16), (700, 7), (1300, 1), (257, 16)])
def test_small_batch_streaming_path_matches_mfma_path(N, T):
    """T <= 16 uses the HBM-bound streaming kernels; it must agree with the MFMA tiles and the oracle
    (Np both an even and an odd multiple of 128)."""
    syn = orc.make_synthetic(3 * N + T, N, 2, 1, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp.set_small_path(True)
    mu_s, var_s = gp.predict(x)
    gp.set_small_path(False)
    mu_m, var_m = gp.predict(x)
    # (Np <= 256 takes the one-launch pass, whose mean is summed in another order: last-bit differences)
    np.testing.assert_allclose(mu_s, mu_m, rtol=1e-12, atol=1e-3 * max(mu_atol(om), 1e-12))
    np.testing.assert_allclose(var_s, var_m, rtol=0, atol=1e-12)
    _, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(var_s, rvar, rtol=0, atol=1e-9)


@pytest.mark.parametrize("N,T,n_s", [(4200, 64, 2), (5000, 48, 2), (5800, 33, 2), (3000, 64, 3), (6500, 64, 3)])
def test_streamed_runs_dealt_to_one_workgroup_per_cu(N, T, n_s):
    """33 .. 64 queries beyond 512 rows: the streamed MFMA kernel's work items are runs of U^-1 rows; when a launch has more
    runs than CUs (N = 5000, two outputs: 258 runs of <= 7 LDS stages) a workgroup takes several, one after the other
    (sr_stream_mfma_kernel<4, true>, planned by sr_stream_items).  Same partial results in the same slots as with one run per
    workgroup: the posterior must agree with the plain MFMA tiles (validated against the oracle at these sizes elsewhere) and
    repeat bit for bit."""
    syn = orc.make_synthetic(7 * N + T, N, n_s, 1, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp.set_small_path(True)
    mu_s, var_s = gp.predict(x)
    mu_2, var_2 = gp.predict(x)
    np.testing.assert_array_equal(mu_s, mu_2)
    np.testing.assert_array_equal(var_s, var_2)
    mu_h, var_h = gp.predict(x[: T // 2 + 1])                 # another width class, other runs: same numbers to rounding
    gp.set_small_path(False)
    mu_m, var_m = gp.predict(x)
    # (the two routes split the sums over the training points differently: SURVEY 8(d)'s 1e-12 sigma_f |alpha|_1)
    scale = float(np.sqrt(np.max(syn["signal_var"])) * np.abs(gp.beta).sum(0).max())
    np.testing.assert_allclose(mu_s, mu_m, rtol=1e-11, atol=1e-12 * scale)
    np.testing.assert_allclose(var_s, var_m, rtol=0, atol=1e-12)
    np.testing.assert_allclose(var_h, var_m[: T // 2 + 1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(mu_h, mu_m[: T // 2 + 1], rtol=1e-11, atol=1e-12 * scale)


@pytest.mark.parametrize("N,T", [(1300, 17), (1300, 300), (2500, 129), (2000, 500), (3100, 1000), (1900, 257), (4100, 130),
                                 (600, 700), (300, 1100), (450, 1400), (512, 2600)])
def test_splitk_path_matches_plain_path(N, T):
    """few query tiles -> the K range of a tile is shared between workgroups: balanced shares of the k-blocks of all tiles
    (sr_var_bal_kernel; (1300, 17) streams instead), the segments of a tile added by a second launch in ascending k order.
    From 256 cells on 256 workgroups take equal shares (some cover whole tiles, the end of one and the start of the next);
    below -- the last three cases: small models with more than 1024 queries -- every k-block is a workgroup of its own (the
    region of the chunked split-K route of rounds 1 - 4, removed in round 5).  Must agree with the plain tiles, and with
    itself call after call."""
    syn = orc.make_synthetic(N + T, N, 2, 1, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp.set_small_path(True)
    mu_s, var_s = gp.predict(x)
    for _ in range(3):
        mu_r, var_r = gp.predict(x)
        np.testing.assert_array_equal(var_r, var_s)          # deterministic: segments are added in ascending k order
    gp.set_small_path(False)
    mu_m, var_m = gp.predict(x)
    # (the mean: the same K* pass on every route -- except where the streamed route evaluates K* inside its MFMA kernel and
    #  adds the mean's partial sums per 128-row chunk, (1300, 17))
    cmp_mu = np.testing.assert_array_equal if T > 32 else (lambda u, v: np.testing.assert_allclose(u, v, rtol=1e-11, atol=1e-12))
    cmp_mu(mu_s, mu_m)
    np.testing.assert_allclose(var_s, var_m, rtol=0, atol=1e-12)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    _, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(var_s, rvar, rtol=0, atol=1e-9)


def _kern_hyp(g, kt, n_out=2):
    hyp = []
    for d in range(n_out):
        pref = "hyp%d_" % d
        hyp.append({k[len(pref):]: g[k] for k in g.files if k.startswith(pref)})
    return hyp


@pytest.mark.parametrize("kt", ["mat52", "lin_rbf", "lin_mat52"])
def test_non_rbf_kernels_vs_reference_formulas(kt):
    """Matern-5/2 and the linear x stationary + linear kernels.  Fixtures hold the
    outputs of the reference's own _k_mat52/_k_lin/_k_lin_rbf/_k_lin_mat52 + gp_pred on numbers."""
    from safe_exploration_amd import SimpleGPModel
    g = load_golden("kern_%s.npz" % kt)
    hyp = _kern_hyp(g, kt)
    for h, nv in zip(hyp, g["noise_var"]):
        h["noise_variance"] = nv - 1e-5
    gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hyp)
    gp.train(g["Z"], g["Y"], opt_hyp=False)
    np.testing.assert_allclose(gp.beta, g["beta"], rtol=1e-7, atol=1e-9 * np.abs(g["beta"]).max())
    mu, var, jac = gp.predict(g["x_new"], None, True)
    scale = np.abs(g["beta"]).sum(0).max()
    np.testing.assert_allclose(mu, g["ref_mu"], rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(var, g["ref_var"], rtol=0, atol=1e-8 * max(1.0, float(g["ref_var"].max())))
    np.testing.assert_allclose(jac, g["jac_fd"], rtol=2e-5, atol=1e-6 * scale)    # central differences (fixture)
    # analytic Jacobian of the oracle (validated against torch-fp64 autograd on the CPU): same bar as the RBF path
    beta_o, _ = orc.gp_fit_k(g["Z"], g["Y"], [kt] * 2, hyp, g["noise_var"])
    np.testing.assert_allclose(jac, orc.gp_mean_jacobian_k(g["x_new"], g["Z"], beta_o, [kt] * 2, hyp), rtol=1e-9,
                               atol=1e-11 * scale)
    # the Gram matrix the factorisation saw: K^-1 from the device vs the reference's kernel matrix
    np.testing.assert_allclose(orc.kernel_matrix(kt, hyp[0], g["x_new"], g["Z"]), g["ref_kstar0"], rtol=1e-12, atol=1e-14)
    Ky = orc.kernel_matrix(kt, hyp[0], g["Z"], g["Z"]) + (g["noise_var"][0] + 1e-8) * np.eye(g["Z"].shape[0])
    np.testing.assert_allclose(gp.inv_K[0].dot(Ky), np.eye(Ky.shape[0]), rtol=0, atol=1e-6)
    m1, s1, j1 = gp(g["x_new"][2:3, :2], g["x_new"][2:3, 2:])
    np.testing.assert_allclose(m1[:, 0], g["ref_mu"][2], rtol=1e-9, atol=1e-11 * scale)
    # second-order outputs of the CasADi Jacobian callback for this kernel (closed forms; the reference leaves
    # them to CasADi's AD): oracle == hand-differentiated formulas checked against finite differences on the CPU
    beta, inv_K = orc.gp_fit_k(g["Z"], g["Y"], [kt] * 2, hyp, g["noise_var"])
    for q in (0, 3):
        x1 = g["x_new"][q]
        rjv, rhm = orc.gp_linearize_extras_k(x1, g["Z"], beta, inv_K, [kt] * 2, hyp)
        mu1, var1, jm1, jv1, hm1 = gp.linearize_predict(x1[None, :2], x1[None, 2:], True)
        np.testing.assert_allclose(mu1[:, 0], mu[q], rtol=1e-9, atol=1e-11 * scale)
        np.testing.assert_allclose(jm1, jac[q], rtol=1e-9, atol=1e-10 * scale)
        np.testing.assert_allclose(jv1, rjv, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(rjv).max()))
        np.testing.assert_allclose(hm1, rhm, rtol=1e-8, atol=1e-10 * scale)
        np.testing.assert_array_equal(hm1, np.transpose(hm1, (0, 2, 1)))


def test_reachability_with_lin_mat52_kernel():
    """the journal experiments' kernel through the fused path == predict + ellipsoid kernel, and the
    oracle's algebra on the same GP outputs."""
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach
    g = load_golden("kern_lin_mat52.npz")
    hyp = _kern_hyp(g, "lin_mat52")
    for h, nv in zip(hyp, g["noise_var"]):
        h["noise_variance"] = nv - 1e-5
    gp = SimpleGPModel(2, 2, 1, kern_types=["lin_mat52"] * 2, hyp=hyp)
    gp.train(g["Z"], g["Y"], opt_hyp=False)
    T = g["x_new"].shape[0]
    rng = np.random.default_rng(8)
    p, kff = g["x_new"][:, :2], g["x_new"][:, 2:]
    kfb = 0.1 * rng.standard_normal((T, 1, 2))
    A = rng.standard_normal((T, 2, 2))
    Q = 0.01 * np.einsum('tij,tkj->tik', A, A) + 0.01 * np.eye(2)[None]
    l = np.array([0.05, 0.02])
    p1, q1 = reach.onestep_reachability_batch(p, gp, kff, l, l, Q, kfb, 2.0, check_bounds=True)
    mu, var, jac = gp.predict(g["x_new"], None, True)
    p2, q2 = reach.ellipsoid_step_batch(p, kff, mu, var, jac, l, l, Q, kfb, 2.0)
    np.testing.assert_allclose(p1, p2, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(q1, q2, rtol=1e-12, atol=1e-16)
    for t in range(T):
        rp, rq = orc.onestep_reachability_from_gp(p[t], Q[t], kff[t], kfb[t], mu[t], var[t], jac[t], l, l, 2.0,
                                                  np.eye(2), np.zeros((2, 1)))
        np.testing.assert_allclose(q1[t], rq, rtol=1e-11, atol=1e-15)


def test_full_size_headline_config():
    """The headline configuration itself (C2': N=5000, T=65536): a 2048-query sample against the oracle
    (explicit-inverse route, factorised on the CPU) plus positivity / symmetry of every result."""
    from safe_exploration_amd import gp_reachability as reach
    N, T = 5000, 65536
    syn = orc.make_synthetic(5, N, 2, 1, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    l = np.array([0.05, 0.02])
    p1, q1, var = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0,
                                                   return_var=True)
    assert np.all(np.isfinite(q1)) and var.min() > 0 and var.max() <= 1.0 + 1e-12
    assert np.linalg.eigvalsh(q1).min() > 0
    om = cached_oracle_model(5, N, 2, 1)
    idx = np.random.default_rng(0).choice(T, 2048, replace=False)
    rp, rq, rvar = orc.onestep_reachability_vectorised(om, syn["p"][idx], syn["Q"][idx], syn["k_ff"][idx],
                                                       syn["k_fb"][idx], l, l, 2.0, np.eye(2), np.zeros((2, 1)))
    np.testing.assert_allclose(p1[idx], rp, rtol=1e-9, atol=max(mu_atol(om), 1e-12))
    np.testing.assert_allclose(var[idx], rvar, rtol=0, atol=1e-9)        # 1e-9 * sigma_f^2
    # Q1 inherits the variance difference through n_s c^2 (sigma+u)/sigma (1+1/c2) <~ 30:
    # cond(K_y) ~ N sf2/sn2 = 5e5 makes the explicit-inverse oracle itself uncertain at the 1e-10 level
    np.testing.assert_allclose(q1[idx], rq, rtol=1e-8, atol=30 * 1e-9)
    # the oracle's second (triangular-solve) route agrees with the HIP factor route far more tightly
    x = np.hstack((syn["p"][idx[:256]], syn["k_ff"][idx[:256]]))
    _, cvar = orc.gp_predict_chol(x, om["Z"], om["beta"], om["chol"], om["lengthscale"], om["signal_var"])
    np.testing.assert_allclose(var[idx[:256]], cvar, rtol=0, atol=2e-11)


@pytest.mark.parametrize("name,n_s,n_u", [("scen_invpend.npz", 2, 1), ("scen_cartpole.npz", 4, 1)])
def test_reference_test_scenarios_on_reference_data(name, n_s, n_u):
    """the reference tests' canonical scenarios (its own data files, seed 125, c_safety 2, L = 0.001) through
    the reference-compatible single-query API; expected values come from the imported reference functions."""
    from safe_exploration_amd import gp_reachability as reach
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    for tag, a, b in (("id", None, None), ("lin", g["a_lin"], g["b_lin"])):
        p1, q1 = reach.onestep_reachability(g["p"], gp, g["k_ff"], g["L"], g["L"], g["q0"], g["k_fb"], 2, 0, a=a, b=b)
        np.testing.assert_allclose(p1, g["p1_ell_" + tag], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(q1, g["q1_ell_" + tag], rtol=1e-8)
        p1, q1 = reach.onestep_reachability(g["p"], gp, g["k_ff"], g["L"], g["L"], None, g["k_fb"], 2, 0, a=a, b=b)
        np.testing.assert_allclose(p1, g["p1_pt_" + tag], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(q1, g["q1_pt_" + tag], rtol=1e-8, atol=1e-14)
        _, _, pa, qa = reach.multistep_reachability(g["p"], gp, g["k_fb_apply"], g["k_ff_all"], g["L"], g["L"], None,
                                                    2, 0, a, b, None)
        np.testing.assert_allclose(pa, g["ms_p_" + tag], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(qa, g["ms_q_" + tag], rtol=1e-6)


def test_models_that_change_size_reuse_device_blocks_and_streams():
    """A model whose N changes from call to call (the exploration loop: update_model after every episode) gets a new
    handle each time; its big buffers come from the library's block cache (zeroed on re-use) and its update runs on the
    process-wide streams.  Same posterior as a model fitted once at that size; release_cached_memory() in between."""
    import safe_exploration_amd as sea
    syn = orc.make_synthetic(12, 1500, 2, 1, 32)
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp = hip_model(syn["Z"][:900], syn["Y"][:900], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    for k, n in enumerate((1300, 700, 1500, 1000, 1300, 900)):
        gp.train(syn["Z"][:n], syn["Y"][:n], opt_hyp=False)
        mu, var = gp.predict(x)
        fresh = hip_model(syn["Z"][:n], syn["Y"][:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
        mu_f, var_f = fresh.predict(x)
        np.testing.assert_array_equal(mu, mu_f)
        np.testing.assert_array_equal(var, var_f)
        del fresh
        if k == 2:
            sea.release_cached_memory()
    # the size policy: few new points are appended to the factor, many refactorise (N / 5, at least 16)
    gp.train(syn["Z"][:1000], syn["Y"][:1000], opt_hyp=False)
    h0 = gp._handle
    gp.update_model(syn["Z"][1000:1100], syn["Y"][1000:1100], opt_hyp=False, replace_old=False)
    assert gp._handle is h0 and gp._handle.N == 1100              # appended in place
    gp.update_model(syn["Z"][1100:1400], syn["Y"][1100:1400], opt_hyp=False, replace_old=False)
    assert gp._handle is not h0 and gp._handle.N == 1400          # 300 > 1100 / 5: a refit (new handle)
    # a chain of appends across several padded sizes on the same model, against one fit of everything
    gp.train(syn["Z"][:1000], syn["Y"][:1000], opt_hyp=False)
    for lo in range(1000, 1500, 100):
        gp.update_model(syn["Z"][lo:lo + 100], syn["Y"][lo:lo + 100], opt_hyp=False, replace_old=False)
    full = hip_model(syn["Z"][:1500], syn["Y"][:1500], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    mu, var = gp.predict(x)
    mu_f, var_f = full.predict(x)
    np.testing.assert_allclose(mu, mu_f, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, var_f, rtol=0, atol=1e-10)


@pytest.mark.parametrize("N0,adds", [(100, [1]), (120, [8, 1]), (250, [6, 128, 3]), (384, [130]), (700, [40]),
                                     (300, [16, 16, 1, 5, 2]), (127, [1, 1, 1]), (1300, [2, 16, 1]), (1, [1, 3])])
def test_row_append_update_equals_refit(N0, adds):
    """update_model(replace_old=False): block row append of the factor == refactorising on all the data
    (SURVEY 8(f).3); crosses 128-padding boundaries, m = 1, m = 128 and m > 128 (two chunks)."""
    ntot = N0 + sum(adds)
    syn = orc.make_synthetic(ntot, ntot, 2, 1, 64)
    Z, Y = syn["Z"], syn["Y"]
    gp = hip_model(Z[:N0], Y[:N0], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    gp.append_limit = 10 ** 9        # the mechanism under test, whatever the size policy (N / 5) would choose
    lo = N0
    for m in adds:
        gp.update_model(Z[lo:lo + m], Y[lo:lo + m], opt_hyp=False, replace_old=False)
        lo += m
    assert gp.x_train.shape[0] == ntot and gp._handle.N == ntot
    full = hip_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu_a, var_a, jac_a = gp.predict(x, None, True)
    mu_f, var_f, jac_f = full.predict(x, None, True)
    scale = np.abs(full.beta).sum(0).max()
    np.testing.assert_allclose(gp.beta, full.beta, rtol=1e-7, atol=1e-9 * np.abs(full.beta).max())
    np.testing.assert_allclose(mu_a, mu_f, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(jac_a, jac_f, rtol=1e-9, atol=1e-10 * scale)
    np.testing.assert_allclose(var_a, var_f, rtol=0, atol=1e-9)
    om = oracle_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    _, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(var_a, rvar, rtol=0, atol=1e-9)
    # single-query latency path and explicit inverse after an append
    np.testing.assert_allclose(gp.predict(x[:1])[1], var_f[:1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.inv_K[0], om["inv_K"][0], rtol=1e-6, atol=1e-8 * np.abs(om["inv_K"][0]).max())
    # the factor itself is complete after any sequence of appends (the one-pass move of small appends relies on the
    # zeros below the diagonal and on the identity padding of the buffer it writes into): equal to the refit's, entry
    # by entry, lower triangle and padding included
    _, wt_a = gp.export_state()
    _, wt_f = full.export_state()
    wa, wf = wt_a.cpu().numpy(), wt_f.cpu().numpy()
    off = gp._handle.Np - ntot
    for d in range(2):
        assert np.all(np.tril(wa[d], -1) == 0.0)
        assert np.all(wa[d][:off, :off] == np.eye(off)) and np.all(wa[d][:off, off:] == 0.0)
        np.testing.assert_allclose(wa[d], wf[d], rtol=1e-6, atol=1e-9 * np.abs(wf[d]).max())


@pytest.mark.parametrize("N0,n_s,n_u,steps", [(1, 2, 1, 5), (60, 4, 1, 6), (126, 2, 1, 4), (200, 3, 2, 3), (254, 2, 1, 5),
                                               (256, 4, 1, 2), (300, 2, 1, 3), (383, 4, 1, 3), (510, 2, 1, 4),
                                               (600, 2, 1, 3), (1022, 2, 1, 4), (1151, 4, 1, 2), (2047, 3, 2, 2)])
def test_one_point_appends_to_small_models_in_one_launch(N0, n_s, n_u, steps):
    """ONE new point is one launch -- one workgroup per output and share of the rows up to 512 padded rows
    (sr_append1_small_kernel), a grid of workgroups with two device-wide barriers up to 8192 (sr_append1_grid_kernel):
    against the refit on all the data (factor entry by entry, zeros and identity padding included), against the general
    route of the same library (set_small_path(0)), the oracle's variance, the log determinant of both routes; crosses the
    padded sizes 128 -> .. -> 640 (from 640 rows on the grid kernel), 1024 -> 1152 (an odd number of 128-row chunks)
    -> 1280, 2048 -> 2176; a point that breaks the factorisation down leaves the model as it was."""
    import ctypes
    from safe_exploration_amd._lib import lib
    ntot = N0 + steps
    syn = orc.make_synthetic(500 + N0, ntot, n_s, n_u, 32)
    Z, Y = syn["Z"], syn["Y"]
    gps = []
    for mode in (1, 0):
        gp = hip_model(Z[:N0], Y[:N0], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
        gp.append_limit = 10 ** 9
        gp.set_small_path(mode)
        for i in range(N0, ntot):
            gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
            assert gp._handle.N == i + 1
        gp.set_small_path(1)
        gps.append(gp)
    full = hip_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    wf = full.export_state()[1].cpu().numpy()
    off = full._handle.Np - ntot
    lds = []
    for gp in gps:
        wa = gp.export_state()[1].cpu().numpy()
        for d in range(n_s):
            assert np.all(np.tril(wa[d], -1) == 0.0)
            assert np.all(wa[d][:off, :off] == np.eye(off)) and np.all(wa[d][:off, off:] == 0.0)
            np.testing.assert_allclose(wa[d], wf[d], rtol=1e-6, atol=1e-9 * np.abs(wf[d]).max())
        np.testing.assert_allclose(gp.beta, full.beta, rtol=1e-7, atol=1e-9 * np.abs(full.beta).max())
        np.testing.assert_array_equal(gp.y_train, Y)
        host = (ctypes.c_double * n_s)()
        assert lib.sr_gp_logdet_cached(gp._handle.h, host) == 0
        lds.append(np.array(host[:]))
    np.testing.assert_allclose(lds[0], lds[1], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(gps[0].information_gain(), full.information_gain(), rtol=1e-10, atol=1e-8)
    x = np.hstack((syn["p"], syn["k_ff"]))
    om = oracle_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    rmu, rvar, _ = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
    mu, var = gps[0].predict(x)
    np.testing.assert_allclose(mu, rmu, rtol=1e-8, atol=max(mu_atol(om), 1e-12) * 10)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    # breakdown: a NaN input gives a Schur complement that is not positive
    gp = gps[0]
    bad = Z[:1].copy()
    bad[0, 0] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        gp.update_model(bad, Y[:1], opt_hyp=False, replace_old=False)
    assert gp._handle.N == ntot and gp.x_train.shape[0] == ntot
    mu2, var2 = gp.predict(x)
    np.testing.assert_array_equal(mu2, mu)
    np.testing.assert_array_equal(var2, var)


def test_in_place_appends_and_what_follows_them():
    """Beyond 512 padded rows a one-point append that keeps the padded size is IN PLACE: the model's buffers become views one
    step further into their allocations (sr_gp::slide).  Everything that may follow such appends: a big batch (tile kernels
    read U^-1 in 16-byte pieces: plain buffers first), an append of several points, more single points, a refit on the same
    handle, an export -- each against a model fitted on the same data from scratch; an odd and an even number of steps."""
    syn = orc.make_synthetic(77, 740, 2, 1, 8)
    Z, Y = syn["Z"], syn["Y"]
    ref = lambda n: hip_model(Z[:n], Y[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    rng = np.random.default_rng(4)
    xq = np.hstack((rng.uniform(-1, 1, (3000, 2)), rng.uniform(-1, 1, (3000, 1))))
    gp = ref(700)
    gp.append_limit = 10 ** 9
    n = 700
    for steps in (5, 2):
        for i in range(n, n + steps):
            gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
        n += steps
        full = ref(n)
        for u, v in zip(gp.export_state(), full.export_state()):                      # alpha, U^-1 (views exported as matrices)
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-6, atol=1e-9 * float(v.abs().max()))
        m1, v1 = gp.predict(xq[:4])                                                   # single-query / small-batch routes on the views
        m2, v2 = full.predict(xq[:4])
        np.testing.assert_allclose(m1, m2, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-10)
        m1, v1 = gp.predict(xq)                                                       # tile kernels
        m2, v2 = full.predict(xq)
        np.testing.assert_allclose(m1, m2, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-10)
    # in-place steps, then several points at once, then a single one again
    for i in range(n, n + 3):
        gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
    gp.update_model(Z[n + 3:n + 9], Y[n + 3:n + 9], opt_hyp=False, replace_old=False)
    gp.update_model(Z[n + 9:n + 10], Y[n + 9:n + 10], opt_hyp=False, replace_old=False)
    n += 10
    full = ref(n)
    m1, v1 = gp.predict(xq[:300])
    m2, v2 = full.predict(xq[:300])
    np.testing.assert_allclose(m1, m2, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-10)
    np.testing.assert_allclose(gp.information_gain(), full.information_gain(), rtol=1e-10, atol=1e-8)
    # a refit of the same handle after in-place steps (the factorisation writes into plain, clean buffers)
    gp.update_model(Z[n:n + 1], Y[n:n + 1], opt_hyp=False, replace_old=False)
    gp.train(Z[:n + 1], Y[:n + 1], opt_hyp=False)
    full = ref(n + 1)
    for u, v in zip(gp.export_state(), full.export_state()):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())


def test_grid_append_that_cannot_assemble_falls_back():
    """The one-launch append beyond 512 padded rows is a grid of workgroups that wait for each other; where the grid cannot
    become resident as a whole, its barrier is given up after ~5 ms, nothing of the model is written, and the append is done
    by separate launches (sr_gp_append1_host says SR_EBUSY -- transient, nothing touched -- and the Python layer stages the
    point for sr_gp_append).  Forced here with sr_test_grid_append_abort; in place (padded size stays) and into new buffers
    (padded size grows); the appends after it work as before.  Round 6: after an abort the library leaves the grid alone for
    the next 16 one-point appends (every further attempt would cost its ~5 ms time-out again) and counts the aborts."""
    import ctypes
    from safe_exploration_amd._lib import lib
    syn = orc.make_synthetic(78, 780, 2, 1, 8)
    Z, Y = syn["Z"], syn["Y"]
    ref = lambda n: hip_model(Z[:n], Y[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    for n0 in (700, 768):                                       # 768: the padded size grows with the next point
        gp = ref(n0)
        gp.append_limit = 10 ** 9
        assert lib.sr_test_grid_append_abort(2) == 0            # the host route gives up; the staged call then rests the grid
        gp.update_model(Z[n0:n0 + 1], Y[n0:n0 + 1], opt_hyp=False, replace_old=False)
        assert lib.sr_test_grid_append_abort(0) == 0
        n_ab = ctypes.c_long(-1)
        assert lib.sr_gp_grid_append_aborts(gp._handle.h, ctypes.byref(n_ab)) == 0 and n_ab.value == 1
        for i in range(n0 + 1, n0 + 4):
            gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
        assert lib.sr_gp_grid_append_aborts(gp._handle.h, ctypes.byref(n_ab)) == 0 and n_ab.value == 1
        full = ref(n0 + 4)
        for u, v in zip(gp.export_state(), full.export_state()):
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-6, atol=1e-9 * float(v.abs().max()))
        m1, v1 = gp.predict(x)
        m2, v2 = full.predict(x)
        np.testing.assert_allclose(m1, m2, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(v1, v2, rtol=0, atol=1e-10)


def test_appends_alternating_with_big_batches():
    """a tile route of the posterior pass reads U^-1 in 16-byte pieces and, after an ODD number of in-place
    one-point appends, first copies the model back into plain buffers (a device-wide wait).  A loop of "one append, one big
    batch" used to pay that every step; now such an unslide keeps the next 64 one-point appends off the in-place route.  The
    numbers must not care: twelve such steps on a model beyond 512 padded rows against a refit, the big batch and a single
    query after every append."""
    syn = orc.make_synthetic(79, 720, 2, 1, 1500)
    Z, Y = syn["Z"], syn["Y"]
    ref = lambda n: hip_model(Z[:n], Y[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    n0 = 700
    gp = ref(n0)
    gp.append_limit = 10 ** 9
    for i in range(n0, n0 + 12):
        gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
        m_big, v_big = gp.predict(x)                          # 1500 queries: the MFMA tiles
        m_one, v_one = gp.predict(x[:1])                      # the streamed single-query kernel on whatever view is current
        np.testing.assert_allclose(m_one, m_big[:1], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(v_one, v_big[:1], rtol=0, atol=1e-10)
        if i in (n0, n0 + 5, n0 + 11):
            full = ref(i + 1)
            m2, v2 = full.predict(x)
            np.testing.assert_allclose(m_big, m2, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(v_big, v2, rtol=0, atol=1e-10)


@pytest.mark.parametrize("kt", ["rbf", "lin_mat52"])
def test_one_point_append_from_host_memory_equals_the_device_pointer_route(kt):
    """sr_gp_append1_host (the new point in the kernel arguments, status words and log det through a pinned block the kernel
    writes) against sr_gp_append with device pointers: the same launch, the same bits -- on both one-launch routes; declined
    (SR_EUNSUPPORTED, nothing touched) where they are switched off."""
    import ctypes
    import torch
    from safe_exploration_amd import _lib, _buffers as B
    syn = orc.make_synthetic(611, 140, 2, 1, 8)
    Z, Y = syn["Z"], syn["Y"]

    def model(n):
        if kt == "rbf":
            return hip_model(Z[:n], Y[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
        from safe_exploration_amd import SimpleGPModel
        rng = np.random.default_rng(5)
        hyp = [dict(orc.make_hyp(kt, rng, 3), noise_variance=nv) for nv in (0.02, 0.03)]
        gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hyp, device="cuda:0")
        gp.train(Z[:n], Y[:n], opt_hyp=False)
        return gp

    a, b = model(120), model(120)
    a.append_limit = b.append_limit = 10 ** 9
    for i in range(120, 135):                                    # crosses 128 -> 256 padded rows
        a.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)          # host route
        hd = b._handle
        tx, ty = B.as_dev(Z[i:i + 1], hd.device), B.as_dev(Y[i:i + 1], hd.device)
        info = (ctypes.c_int * 2)()
        _lib.check(_lib.lib.sr_gp_append(hd.h, B.ptr(tx), B.ptr(ty), 1, B.stream_ptr(hd.device), info))
        torch.cuda.synchronize()
        npad = ctypes.c_long(0)                                  # (what SimpleGPModel._append keeps on the host side)
        _lib.check(_lib.lib.sr_gp_padded_n(hd.h, ctypes.byref(npad)))
        hd.N, hd.Np = hd.N + 1, npad.value
    wa, wb = a.export_state(), b.export_state()
    for u, v in zip(wa, wb):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())
    la, lb = (ctypes.c_double * 2)(), (ctypes.c_double * 2)()
    assert _lib.lib.sr_gp_logdet_cached(a._handle.h, la) == 0 and _lib.lib.sr_gp_logdet_cached(b._handle.h, lb) == 0
    assert list(la) == list(lb)
    # beyond 512 padded rows: the grid kernel, the same two entry points, the same bits; declined where the one-launch
    # routes are switched off (nothing touched)
    bigs = [hip_model(np.vstack([Z] * 5)[:600] + 0.01 * np.arange(600)[:, None], np.vstack([Y] * 5)[:600], syn["lengthscale"],
                      syn["signal_var"], syn["noise_var"], 2, 1) for _ in range(2)]
    x1, y1 = np.ascontiguousarray(Z[0] + 0.5), np.ascontiguousarray(Y[0])
    info = (ctypes.c_int * 2)()
    bigs[0].set_small_path(0)
    rc = _lib.lib.sr_gp_append1_host(bigs[0]._handle.h, ctypes.c_void_p(x1.ctypes.data), ctypes.c_void_p(y1.ctypes.data),
                                     B.stream_ptr(bigs[0]._handle.device), info)
    assert rc == _lib.SR_EUNSUPPORTED
    n = ctypes.c_long(0)
    assert _lib.lib.sr_gp_padded_n(bigs[0]._handle.h, ctypes.byref(n)) == 0 and n.value == 640
    bigs[0].set_small_path(1)
    _lib.check(_lib.lib.sr_gp_append1_host(bigs[0]._handle.h, ctypes.c_void_p(x1.ctypes.data), ctypes.c_void_p(y1.ctypes.data),
                                           B.stream_ptr(bigs[0]._handle.device), info))
    hd = bigs[1]._handle
    tx, ty = B.as_dev(x1[None, :], hd.device), B.as_dev(y1[None, :], hd.device)
    _lib.check(_lib.lib.sr_gp_append(hd.h, B.ptr(tx), B.ptr(ty), 1, B.stream_ptr(hd.device), info))
    torch.cuda.synchronize()
    for g_ in bigs:
        g_._handle.N = 601
    for u, v in zip(bigs[0].export_state(), bigs[1].export_state()):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())


@pytest.mark.parametrize("kt,N0,adds", [("lin_mat52", 90, [1, 4]), ("mat52", 250, [16, 1]), ("lin_rbf", 600, [3, 1, 40]),
                                        ("lin_mat52", 125, [1] * 6), ("mat52", 30, [1, 1, 1])])
def test_row_append_with_the_journal_kernels(kt, N0, adds):
    """update_model(replace_old=False) on models with the kernels of the reference's journal experiments (Matern-5/2,
    linear x stationary + linear): the short appends (all outputs per launch, the Schur complement's Gram block from
    sr_gram_general_kernel) and the 17..128-row route against a refit on all the data."""
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(77 + N0)
    D, ntot = 3, N0 + sum(adds)
    Z = rng.uniform(-1, 1, (ntot, D))
    Y = rng.standard_normal((ntot, 2))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(2)]
    noise = np.array([0.02, 0.03])
    hh = [dict(h, noise_variance=nv) for h, nv in zip(hyp, noise)]
    gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hh)
    gp.train(Z[:N0], Y[:N0], opt_hyp=False)
    gp.append_limit = 10 ** 9
    lo = N0
    for m in adds:
        gp.update_model(Z[lo:lo + m], Y[lo:lo + m], opt_hyp=False, replace_old=False)
        lo += m
    assert gp._handle.N == ntot
    full = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hh)
    full.train(Z, Y, opt_hyp=False)
    np.testing.assert_allclose(gp.beta, full.beta, rtol=1e-6, atol=1e-8 * np.abs(full.beta).max())
    x = rng.uniform(-0.8, 0.8, (40, D))
    mu_a, var_a, jac_a = gp.predict(x, None, True)
    mu_f, var_f, jac_f = full.predict(x, None, True)
    scale = max(np.abs(full.beta).sum(0).max(), 1.0)
    np.testing.assert_allclose(mu_a, mu_f, rtol=1e-8, atol=1e-10 * scale)
    np.testing.assert_allclose(jac_a, jac_f, rtol=1e-8, atol=1e-9 * scale)
    np.testing.assert_allclose(var_a, var_f, rtol=0, atol=1e-9 * scale)
    wa, wf = gp.export_state()[1].cpu().numpy(), full.export_state()[1].cpu().numpy()
    for d in range(2):
        assert np.all(np.tril(wa[d], -1) == 0.0)
        np.testing.assert_allclose(wa[d], wf[d], rtol=1e-6, atol=1e-9 * np.abs(wf[d]).max())
    beta_ref, inv_K = orc.gp_fit_k(Z, Y, [kt] * 2, hyp, noise + 1e-5)
    rmu, rvar = orc.gp_predict_k(x, Z, beta_ref, inv_K, [kt] * 2, hyp)
    np.testing.assert_allclose(mu_a, rmu, rtol=1e-7, atol=1e-8 * scale)
    np.testing.assert_allclose(var_a, rvar, rtol=0, atol=1e-7 * scale)


def test_small_appends_after_refit_and_release():
    """The buffer a small append writes into may hold an older state of the model (ping-pong), a released or a fresh
    allocation: always a complete factor afterwards."""
    syn = orc.make_synthetic(321, 420, 2, 1, 16)
    Z, Y = syn["Z"], syn["Y"]
    gp = hip_model(Z[:400], Y[:400], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    gp.update_model(Z[400:403], Y[400:403], opt_hyp=False, replace_old=False)      # fresh buffer
    gp.update_model(Z[403:404], Y[403:404], opt_hyp=False, replace_old=False)      # ping
    gp.update_model(Z[404:410], Y[404:410], opt_hyp=False, replace_old=False)      # pong
    gp.release_scratch()
    gp.update_model(Z[410:411], Y[410:411], opt_hyp=False, replace_old=False)      # fresh again
    # an append that breaks down (NaN inputs: the Schur complement is not positive) leaves the model as it was ...
    before = gp.predict(np.hstack((syn["p"], syn["k_ff"])))
    bad = Z[411:415].copy()
    bad[2, 0] = np.nan
    with pytest.raises(np.linalg.LinAlgError):
        gp.update_model(bad, Y[411:415], opt_hyp=False, replace_old=False)
    assert gp._handle.N == 411 and gp.x_train.shape[0] == 411
    after = gp.predict(np.hstack((syn["p"], syn["k_ff"])))
    np.testing.assert_array_equal(before[0], after[0])
    np.testing.assert_array_equal(before[1], after[1])
    # ... and the next (smaller) one does not inherit the half-written spare buffer
    gp.update_model(Z[411:413], Y[411:413], opt_hyp=False, replace_old=False)
    gp.update_model(Z[413:420], Y[413:420], opt_hyp=False, replace_old=False)
    full = hip_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    wa, wf = gp.export_state()[1].cpu().numpy(), full.export_state()[1].cpu().numpy()
    for d in range(2):
        assert np.all(np.tril(wa[d], -1) == 0.0)
        np.testing.assert_allclose(wa[d], wf[d], rtol=1e-6, atol=1e-9 * np.abs(wf[d]).max())
    x = np.hstack((syn["p"], syn["k_ff"]))
    np.testing.assert_allclose(gp.predict(x)[1], full.predict(x)[1], rtol=0, atol=1e-9)


def test_distance_to_center_batch_vs_reference_golden():
    """A9 batch kernel against the reference's utils_ellipsoid.distance_to_center outputs."""
    from safe_exploration_amd import utils_ellipsoid as ue
    g = load_golden("ellipsoid.npz")
    for n in (2, 3, 4, 8):
        p = g["sum_p1_%d" % n].T
        q = g["sum_q1_%d" % n][None]
        d = ue.distance_to_center_batch(g["dist_s_%d" % n], p, q)
        np.testing.assert_allclose(d[0], g["dist_d_%d" % n], rtol=1e-11)
        inside = ue.sample_inside_ellipsoid_batch(g["dist_s_%d" % n], p, q, 3.0)
        assert list(inside[0]) == list(g["inside_%d" % n])
    # many ellipsoids x per-ellipsoid samples == the host helper applied one by one
    rng = np.random.default_rng(4)
    T, K, n = 50, 9, 4
    A = rng.standard_normal((T, n, n))
    Q = np.einsum('tij,tkj->tik', A, A) + 0.1 * np.eye(n)
    P = rng.standard_normal((T, n))
    S = rng.standard_normal((T, K, n))
    d = ue.distance_to_center_batch(S, P, Q)
    for t in (0, 17, 49):
        np.testing.assert_allclose(d[t], ue.distance_to_center(S[t], P[t][:, None], Q[t]), rtol=1e-10)
    # box corners lie on the covering ellipsoid (reference test_utils_ellipsoid.py:13-25)
    ub = np.array([0.1, 0.2, 0.3])
    corners = np.array([[sx * ub[0], sy * ub[1], sz * ub[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    dc = ue.distance_to_center_batch(corners, np.zeros((1, 3)), ue.ellipsoid_from_rectangle(ub)[None])
    np.testing.assert_allclose(dc, 1.0, rtol=1e-12)


@pytest.mark.parametrize("name,n_s,n_u", [("moments_pend.npz", 2, 1), ("moments_cart.npz", 4, 1)])
def test_moment_propagation_vs_reference_golden(name, n_s, n_u):
    """Taylor / mean-equivalent Gaussian moment propagation; expected values are the reference's
    own multi_step_taylor_symbolic / mean_equivalent_multistep evaluated on numbers."""
    from safe_exploration_amd import uncertainty_propagation_casadi as prop
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    for tag, mode in (("taylor", prop.TAYLOR), ("meaneq", prop.MEAN_EQUIVALENT)):
        mu, sig, gv = prop.multistep_moments_batch(g["mu0"], gp, g["k_ff"], g["k_fb"], g["a_lin"], g["b_lin"], mode)
        np.testing.assert_allclose(mu, g["mu_" + tag], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(sig, g["sigma_" + tag], rtol=1e-7, atol=1e-13)
        if tag == "meaneq":
            np.testing.assert_allclose(gv, g["gpvar_meaneq"], rtol=0, atol=1e-9)
    # reference-shaped single-trajectory API
    mu_all, sigma_all, _ = prop.multi_step_taylor(g["mu0"][0][:, None], gp, g["k_ff"][0], list(g["k_fb"][0]), None,
                                                  g["a_lin"], g["b_lin"])
    H = g["k_ff"].shape[1]
    assert mu_all.shape == (H, n_s) and sigma_all.shape == (H, n_s * n_s)
    np.testing.assert_allclose(sigma_all.reshape(H, n_s, n_s), g["sigma_taylor"][0], rtol=1e-7, atol=1e-13)
    mu_me, sig_me, _ = prop.mean_equivalent_multistep(g["mu0"][1][:, None], gp, g["k_ff"][1], list(g["k_fb"][1]),
                                                      None, g["a_lin"], g["b_lin"])
    np.testing.assert_allclose(mu_me, g["mu_meaneq"][1], rtol=1e-8, atol=1e-11)
    with pytest.raises(NotImplementedError):       # sigma_0 is "Still need to do this" in the reference too (:124)
        prop.multi_step_taylor(g["mu0"][0][:, None], gp, g["k_ff"][0], list(g["k_fb"][0]), np.eye(n_s))
    # one step with a foreign StateSpaceModel (GP outputs computed on the host by the caller)
    om = oracle_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])

    def ssm(s, a):
        m_, v_, j_ = orc._predict_one(om, np.hstack((s, a))[0])
        return m_[:, None], v_[:, None], j_
    m1, s1, _ = prop.one_step_taylor(g["mu_taylor"][0, 0][:, None], ssm, g["k_ff"][0, 1][:, None],
                                     g["sigma_taylor"][0, 0], g["k_fb"][0, 0], g["a_lin"], g["b_lin"])
    np.testing.assert_allclose(m1[:, 0], g["mu_taylor"][0, 1], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(s1, g["sigma_taylor"][0, 1], rtol=1e-10, atol=1e-15)


def test_c_abi_from_plain_hip_program(lib_built):
    """the boundary has no Python/torch dependency: examples/capi_demo.cpp (plain HIP host code) links
    libsafereach.so, reproduces the SURVEY anchor through sr_ellipsoid_step and runs the handle API."""
    import os
    import shutil
    import subprocess
    import tempfile
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = os.path.join(tempfile.mkdtemp(), "capi_demo")
    libdir = os.path.join(ROOT, "safe_exploration_amd")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "capi_demo.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lsafereach", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "capi_demo OK" in res.stdout


def test_rccl_replication_from_plain_hip_program(lib_built):
    """libsafereach_comm.so (sr_comm_init_all / sr_comm_bcast / sr_comm_destroy): examples/comm_demo.cpp factorises on
    device 0, broadcasts alpha and U^-1 over RCCL to every other visible device of the process, shards the queries and
    compares the gathered result with device 0's.  On a one-GPU box the communicator has a single rank."""
    import os
    import shutil
    import subprocess
    import tempfile
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "safe_exploration_amd")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    assert os.path.exists(os.path.join(libdir, "libsafereach_comm.so"))
    exe = os.path.join(tempfile.mkdtemp(), "comm_demo")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "comm_demo.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-lsafereach", "-lsafereach_comm", "-Wl,-rpath," + libdir,
           "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "comm_demo OK" in res.stdout


@pytest.mark.parametrize("n_s,n_u", [(1, 1), (1, 3), (3, 1), (5, 4), (6, 2), (7, 1), (8, 3), (8, 4)])
def test_all_state_action_dimensions(n_s, n_u):
    """every (n_s, n_u) template instance of the ellipsoid kernel (Jacobi eigen-solver for n_s >= 3) and the
    D = 2..12 paths of the covariance kernel: fused one-step + 3-step chain against the oracle."""
    from safe_exploration_amd import gp_reachability as reach
    T, N = 9, 70
    syn = orc.make_synthetic(100 * n_s + n_u, N, n_s, n_u, T, sf2=0.05)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    rng = np.random.default_rng(n_s * 10 + n_u)
    l_mu, l_sigma = rng.uniform(0.01, 0.05, n_s), rng.uniform(0.01, 0.05, n_s)
    a = 0.7 * np.eye(n_s) + 0.05 * rng.standard_normal((n_s, n_s))
    b = 0.1 * rng.standard_normal((n_s, n_u))
    p1, q1, var = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l_mu, l_sigma, syn["Q"], syn["k_fb"],
                                                   1.7, a, b, check_bounds=True, return_var=True)
    rp, rq, rvar = orc.onestep_reachability_batch(om, syn["p"], syn["Q"], syn["k_ff"], syn["k_fb"], l_mu, l_sigma,
                                                  1.7, a, b)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    np.testing.assert_allclose(p1, rp, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(q1, rq, rtol=1e-8, atol=1e-12)
    H = 3
    kfb = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
    kff = 0.1 * rng.standard_normal((T, H, n_u))
    pa, qa = reach.multistep_reachability_batch(syn["p"], gp, kfb, kff, l_mu, l_sigma, None, 1.7, a, b)
    rpa, rqa = orc.multistep_reachability_batch(om, syn["p"], kfb, kff, l_mu, l_sigma, None, 1.7, a, b)
    np.testing.assert_allclose(pa, rpa, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(qa, rqa, rtol=1e-7, atol=1e-11)
    # remainder over-approximation and safety distance for the same dimensions
    from safe_exploration_amd import utils
    um, us = utils.compute_remainder_overapproximations_batch(syn["Q"], syn["k_fb"], l_mu, l_sigma)
    for t in range(T):
        rum, rus = orc.compute_remainder_overapproximations(syn["Q"][t], syn["k_fb"][t], l_mu, l_sigma)
        np.testing.assert_allclose(um[t], rum, rtol=1e-11)
        np.testing.assert_allclose(us[t], rus, rtol=1e-11)


def test_entry_points_are_graph_capturable():
    """steady-state calls allocate nothing through HIP and never synchronise, so the whole H-step chain can be
    captured into a hipGraph (torch.cuda.graph) and replayed on new inputs written into the static buffers."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, workload, _buffers as B
    syn = orc.make_synthetic(31, 200, 2, 1, 8, sf2=0.01)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    dev = gp.device
    roll = workload.random_rollout_controls(3, 256, 6, 2, 1)
    tr = {k: B.as_dev(v, dev) for k, v in roll.items()}
    l, a, b = np.array([0.05, 0.02]), 0.8 * np.eye(2), np.zeros((2, 1))
    f = lambda: reach.multistep_reachability_batch(tr["p0"], gp, tr["k_fb"], tr["k_ff"], l, l, None, 2.0, a, b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            f()                                   # warm-up: workspace + constant caches
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = f()
    roll2 = workload.random_rollout_controls(4, 256, 6, 2, 1)
    for k in tr:
        tr[k].copy_(B.as_dev(roll2[k], dev))      # new inputs into the captured buffers
    g.replay()
    torch.cuda.synchronize()
    got_p, got_q = out[0].clone(), out[1].clone()
    ref_p, ref_q = f()
    assert torch.equal(got_p, ref_p) and torch.equal(got_q, ref_q)
    # replayed again and again (eager calls in between): the hand-off state of the persistent kernel lives on the
    # device, a captured launch carries nothing that goes stale
    assert gp.last_chain
    for seed in (5, 6, 7):
        roll3 = workload.random_rollout_controls(seed, 256, 6, 2, 1)
        for k in tr:
            tr[k].copy_(B.as_dev(roll3[k], dev))
        g.replay()
        torch.cuda.synchronize()
        got_p, got_q = out[0].clone(), out[1].clone()
        ref_p, ref_q = f()
        assert torch.equal(got_p, ref_p) and torch.equal(got_q, ref_q)
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out[0], ref_p) and torch.equal(out[1], ref_q)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_s,n_u", [("mc_pend.npz", 2, 1), ("mc_cart.npz", 4, 1)])
def test_monte_carlo_verification_vs_reference_golden(name, n_s, n_u):
    """MonteCarloSafetyVerification (sampling_models.py:14-107).  Expected particles come from the
    reference's own sample_n_step replaying stored standard-normal draws; here each step is one batched GP
    evaluation of all particles + sr_gp_sample on the device."""
    from safe_exploration_amd.sampling_models import MonteCarloSafetyVerification
    g = load_golden(name)
    gp = hip_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"], n_s, n_u)
    mc = MonteCarloSafetyVerification(gp)
    n, n_samples, _ = g["eps"].shape
    S, S_all = mc.sample_n_step(g["x0"], g["K"], g["k"], n, n_samples, eps=g["eps"])
    assert S_all.shape == (n, n_samples, n_s) and S.shape == (n_samples, n_s)
    np.testing.assert_allclose(S_all, g["S_all"], rtol=1e-8, atol=1e-10)
    ratio, inside = mc.inside_ellipsoid_ratio(g["S_all"], g["ell_q"], g["ell_p"])
    np.testing.assert_array_equal(inside, g["inside"])
    np.testing.assert_allclose(ratio, g["ratio"], rtol=0, atol=1e-15)
    with pytest.raises(AssertionError):
        mc.sample_n_step(g["x0"], g["K"][:, :, :1], g["k"], n, n_samples)


@pytest.mark.gpu
def test_sample_from_gp_and_information_gain():
    """gaussian_process.py:598-634: marginal posterior samples and log det(I + K/sigma_n^2)."""
    import torch
    syn = orc.make_synthetic(91, 300, 2, 1, 40)
    m = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    eps = np.random.default_rng(3).standard_normal((40, 7, 2))
    S = gp.sample_from_gp(x, size=7, eps=eps)
    np.testing.assert_allclose(S, orc.sample_from_gp(m, x, eps), rtol=1e-9, atol=1e-9)
    # device generator: shape, reproducibility, first two moments of 20000 draws at one input
    gen = torch.Generator(device=gp.device); gen.manual_seed(5)
    big = gp.sample_from_gp(x[:1], size=20000, generator=gen)
    gen.manual_seed(5)
    again = gp.sample_from_gp(x[:1], size=20000, generator=gen)
    assert big.shape == (1, 20000, 2) and np.array_equal(big, again)
    mu, var = gp.predict(x[:1])
    assert np.all(np.abs(big[0].mean(0) - mu[0]) < 5 * np.sqrt(var[0] / 20000))
    np.testing.assert_allclose(big[0].var(0), var[0], rtol=0.05)
    # information gain: exact up to the 1e-8 inference jitter carried by the factor (bound N*1e-8/sigma_n^2)
    ig = gp.information_gain()
    ref = orc.information_gain(syn["Z"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    bound = 300 * 1e-8 / syn["noise_var"].min()
    assert len(ig) == 2 and all(0 <= a - b <= bound + 1e-9 for a, b in zip(ig, ref))
    assert gp.information_gain(gp.z) == ig
    with pytest.raises(ValueError):
        gp.information_gain(syn["Z"][:10])


@pytest.mark.gpu
def test_information_gain_from_the_host_copy_follows_the_model():
    """The <= 16-row append reads log det back with its status words (sr_gp_logdet_cached): the information gain of the
    exploration loop then costs no launch.  The host copy must equal what the device route computes on the same
    factor after short appends, and there is none after a fit or a long append."""
    import ctypes
    from safe_exploration_amd import _buffers as B
    from safe_exploration_amd._lib import lib, check
    syn = orc.make_synthetic(77, 260, 2, 1, 4)
    Z, Y = syn["Z"], syn["Y"]
    gp = hip_model(Z[:150], Y[:150], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    gp.append_limit = 10 ** 9
    hd = gp._handle

    def both():
        host = (ctypes.c_double * 2)()
        rc = lib.sr_gp_logdet_cached(hd.h, host)
        dev = B.empty((2,), hd.device)
        check(lib.sr_gp_logdet(hd.h, B.ptr(dev), B.stream_ptr(hd.device)))
        return rc, np.array(host[:]), B.to_numpy(dev)

    rc, host, dev = both()
    assert rc != 0                                          # a fit leaves no host copy (no extra launch on every refit) ...
    ig0 = gp.information_gain()                            # ... the device route serves
    ref0 = orc.information_gain(Z[:150], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    assert all(-1e-9 <= a - b <= 150 * 1e-8 / syn["noise_var"].min() + 1e-9 for a, b in zip(ig0, ref0))
    n = 150
    for m in (1, 3, 16, 1):                                # short appends keep the copy current
        gp.update_model(Z[n:n + m], Y[n:n + m], opt_hyp=False, replace_old=False)
        n += m
        rc, host, dev = both()
        assert rc == 0 and hd.N == n
        np.testing.assert_allclose(host, dev, rtol=1e-13, atol=0)     # (the one-point kernel adds the logs itself)
        ig = gp.information_gain()
        ref = orc.information_gain(Z[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        bound = n * 1e-8 / syn["noise_var"].min()
        assert all(-1e-9 <= a - b <= bound + 1e-9 for a, b in zip(ig, ref))
    gp.update_model(Z[n:n + 40], Y[n:n + 40], opt_hyp=False, replace_old=False)      # 40 rows: the other route
    n += 40
    rc, _, dev = both()
    assert rc != 0 and hd.N == n
    ig = gp.information_gain()                              # falls back to the device route
    ref = orc.information_gain(Z[:n], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    assert all(-1e-9 <= a - b <= n * 1e-8 / syn["noise_var"].min() + 1e-9 for a, b in zip(ig, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("N,n_s,n_u,T", [(1, 2, 1, 3), (2, 2, 1, 8), (100, 2, 1, 1), (128, 4, 1, 9), (129, 2, 1, 17),
                                         (200, 4, 1, 300), (256, 2, 1, 1024), (150, 3, 2, 64), (257, 2, 1, 33),
                                         (384, 4, 1, 16), (400, 2, 1, 250), (512, 3, 2, 200)])
def test_fused_small_model_pass(N, n_s, n_u, T):
    """K0 (sr_small.hip): Np <= 512 and T <= 1024 evaluate the whole posterior in one launch.  Checked against
    the oracle, against the three-kernel pass of the same library, and that it is the path that ran."""
    from safe_exploration_amd import _lib
    syn = orc.make_synthetic(1000 + 3 * N + T, N, n_s, n_u, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))
    gp.prof_reset(); gp.prof_enable(True)
    mu, var, jac = gp.predict(x, None, True)
    mu_only, var_only = gp.predict(x)
    gp.prof_enable(False)
    assert gp.prof_get(_lib.K_SMALL)[1] == 2 and gp.prof_get(_lib.K_VAR)[1] == 0
    np.testing.assert_array_equal(mu_only, mu)
    np.testing.assert_array_equal(var_only, var)
    rmu, rvar, rjac = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
    at = max(mu_atol(om), 1e-12)
    np.testing.assert_allclose(mu, rmu, rtol=1e-9, atol=at)
    np.testing.assert_allclose(jac, rjac, rtol=1e-9, atol=10 * at)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    gp.set_small_path(2)                                  # same library, K1 -> K2m -> K3
    mu3, var3, jac3 = gp.predict(x, None, True)
    gp.set_small_path(1)
    np.testing.assert_allclose(mu, mu3, rtol=1e-12, atol=1e-3 * at)
    np.testing.assert_allclose(jac, jac3, rtol=1e-11, atol=1e-2 * at)
    np.testing.assert_allclose(var, var3, rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_max_variance_data_selection():
    """choose_datapoints_maxvar (gaussian_process.py:280-345): greedy max-variance subset, one batched predict over
    the pool + one row append per round.  Same picks, in the same order, as the oracle refitting from scratch."""
    from safe_exploration_amd import SimpleGPModel
    syn = orc.make_synthetic(123, 300, 2, 1, 4)
    hyp = hyp_from(syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=hyp, m=40)
    init = [3, 57, 111, 160, 201, 250, 299, 8, 77, 140]
    xs, ys, idx = gp.choose_datapoints_maxvar(syn["Z"], syn["Y"], 40, init_idx=init, return_index=True)
    ref = orc.choose_datapoints_maxvar(syn["Z"], syn["Y"], 40, init, syn["lengthscale"], syn["signal_var"],
                                       syn["noise_var"])
    np.testing.assert_array_equal(idx, ref)
    np.testing.assert_array_equal(xs, syn["Z"][ref])
    np.testing.assert_array_equal(ys, syn["Y"][ref])
    # the incrementally conditioned model equals a fresh fit on the chosen rows
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu_inc, var_inc = gp.predict(x)
    om = oracle_model(xs, ys, syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    rmu, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(mu_inc, rmu, rtol=1e-9, atol=max(mu_atol(om), 1e-12))
    np.testing.assert_allclose(var_inc, rvar, rtol=0, atol=1e-9)
    # the reference's entry points: train(m=...) / update_model with the default choose_data=True (k-means seeds)
    np.random.seed(0)
    gp.train(syn["Z"], syn["Y"], m=40, opt_hyp=False)
    assert gp.z.shape == (40, 3) and gp.x_train.shape == (300, 3)
    assert len({tuple(r) for r in gp.z}) == 40 and all(tuple(r) in {tuple(q) for q in syn["Z"]} for r in gp.z)
    gp.update_model(syn["Z"][:50], syn["Y"][:50], opt_hyp=False, replace_old=True)
    assert gp.z.shape == (40, 3)
    with pytest.warns(UserWarning):
        gp.update_model(syn["Z"][:30], syn["Y"][:30], opt_hyp=False, replace_old=True)      # fewer than m: all of them
    assert gp.z.shape == (30, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("kt", ["rbf", "mat52", "lin_rbf", "lin_mat52"])
def test_marginal_likelihood_and_gradient(kt):
    """sr_gp_mll (objective/gradient of train(opt_hyp=True), gaussian_process.py:249-250) against the oracle's
    per-kernel closed forms (themselves checked against finite differences on the CPU)."""
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(8)
    N, D = 150, 3
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, 2))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(2)]
    gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2)           # nothing fixed: every hyper-parameter is free
    for i in range(2):
        gp.hyp[i] = {k: (np.array(v, dtype=float) if np.ndim(v) else float(v)) for k, v in hyp[i].items()}
        gp._noise[i] = 0.03 + 0.02 * i
        nll, g = gp.neg_log_marginal_likelihood(Z, Y, i)
        rnll, rg = orc.gp_nll_grad(Z, Y[:, i], kt, hyp[i], gp._noise[i])
        ref = np.concatenate([np.reshape(rg[k], (-1,)) for k, _ in gp._free_hyp(i)])
        assert abs(nll - rnll) < 1e-9 * abs(rnll)
        np.testing.assert_allclose(g, ref, rtol=1e-7, atol=1e-8 * np.abs(ref).max())


@pytest.mark.gpu
def test_train_with_hyperparameter_optimisation():
    """train(opt_hyp=True): data drawn from a GP with known hyper-parameters; the maximum-likelihood fit lowers the
    nll below its starting value AND below the value at the generating parameters, ends at a stationary point,
    and recovers them to the accuracy 200 points allow.  Keys passed in ``hyp`` stay fixed."""
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(21)
    N, D = 200, 3
    Z = rng.uniform(-2, 2, (N, D))
    true = {"lengthscale": np.array([0.8, 1.5, 1.1]), "variance": 1.7}
    K = orc.rbf_kernel(Z, Z, true["variance"], true["lengthscale"]) + 0.01 * np.eye(N)
    Y = np.linalg.cholesky(K).dot(rng.standard_normal((N, 2)))
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2)
    start = [gp.neg_log_marginal_likelihood(Z, Y, i, with_grad=False)[0] for i in range(2)]
    gp.train(Z, Y, opt_hyp=True)
    for i in range(2):
        nll, g = gp.neg_log_marginal_likelihood(Z, Y, i)
        at_truth = orc.gp_nll_grad(Z, Y[:, i], "rbf", true, 0.01)[0]
        assert nll < start[i] - 10 and nll <= at_truth + 1e-6
        theta = gp._get_free(i)
        assert np.abs(g * theta).max() < 1e-2                       # stationary in log-parameters
        assert 0.5 * 0.01 < gp._noise[i] < 2 * 0.01
        np.testing.assert_allclose(gp.hyp[i]["lengthscale"], true["lengthscale"], rtol=0.35)
        # the oracle's own optimum from the same start agrees
        rn = orc.gp_nll_grad(Z, Y[:, i], "rbf", gp.hyp[i], gp._noise[i])[0]
        assert abs(rn - nll) < 1e-8 * abs(nll)
    mu, var = gp.predict(Z[:5])
    assert np.abs(mu - Y[:5]).max() < 0.5 and var.max() < 0.1
    # fixed keys stay fixed, the rest (variance, noise) moves
    gp2 = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=[{"lengthscale": np.array([1.0, 1.0, 1.0])}] * 2)
    gp2.train(Z, Y, opt_hyp=True)
    for i in range(2):
        np.testing.assert_array_equal(gp2.hyp[i]["lengthscale"], np.ones(3))
        assert gp2.hyp[i]["variance"] != 1.0 and gp2._noise[i] != 1.0


@pytest.mark.gpu
def test_refit_reuses_the_handle_and_deep_copies_keep_their_model():
    """update_model with the same N refactorises in place (no device allocation); a deep copy (what
    get_forward_model_casadi hands to CasADi, state_space_models.py:166) must keep predicting with the model it
    was copied from, whatever happens to the original afterwards (refit or row append)."""
    import copy
    a = orc.make_synthetic(301, 180, 2, 1, 16)
    b = orc.make_synthetic(302, 180, 2, 1, 16)
    gp = hip_model(a["Z"], a["Y"], a["lengthscale"], a["signal_var"], a["noise_var"], 2, 1)
    x = np.hstack((a["p"], a["k_ff"]))
    mu_a, var_a = gp.predict(x)
    h0 = gp._handle
    gp.update_model(b["Z"], b["Y"], opt_hyp=False, replace_old=True)
    assert gp._handle is h0                                     # reused
    fresh = hip_model(b["Z"], b["Y"], a["lengthscale"], a["signal_var"], a["noise_var"], 2, 1)
    mu_b, var_b = gp.predict(x)
    np.testing.assert_array_equal(mu_b, fresh.predict(x)[0])
    np.testing.assert_array_equal(var_b, fresh.predict(x)[1])
    assert np.abs(mu_b - mu_a).max() > 1e-3
    snap = copy.deepcopy(gp)
    gp.update_model(a["Z"], a["Y"], opt_hyp=False, replace_old=True)
    assert gp._handle is not h0 and snap._handle is h0          # shared handle left alone
    np.testing.assert_array_equal(snap.predict(x)[0], mu_b)
    np.testing.assert_array_equal(gp.predict(x)[0], mu_a)
    snap2 = copy.deepcopy(gp)
    gp.update_model(b["Z"][:20], b["Y"][:20], opt_hyp=False, replace_old=False)    # append path must not touch snap2
    assert gp.z.shape[0] == 200 and snap2.z.shape[0] == 180
    np.testing.assert_array_equal(snap2.predict(x)[0], mu_a)
    both = orc.gp_fit(np.vstack((a["Z"], b["Z"][:20])), np.vstack((a["Y"], b["Y"][:20])), a["lengthscale"],
                      a["signal_var"], a["noise_var"])
    rmu, _ = orc.gp_predict(x, np.vstack((a["Z"], b["Z"][:20])), both[0], both[1], a["lengthscale"], a["signal_var"],
                            False)
    np.testing.assert_allclose(gp.predict(x)[0], rmu, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("N,n_s,n_u", [(1, 2, 1), (60, 2, 1), (128, 4, 1), (200, 2, 1), (256, 3, 2), (200, 4, 4),
                                       (300, 2, 1), (384, 4, 1), (500, 2, 1)])
def test_fused_small_model_linearize(N, n_s, n_u):
    """Np <= 512: linearize_predict(jacobians=True) is ONE launch (the MFMA columns carry [k*, dk*/dx]); checked
    against the oracle's closed forms and against the multi-kernel route of the same library."""
    from safe_exploration_amd import _lib
    syn = orc.make_synthetic(4000 + N + n_u, N, n_s, n_u, 3)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))[1]
    gp.prof_reset(); gp.prof_enable(True)
    mu, var, jm, jv, hm = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
    gp.prof_enable(False)
    if gp._handle.Np < 512:            # (Np = 512 takes the streamed route: measured faster there)
        assert gp.prof_get(_lib.K_SMALL)[1] == 1 and gp.prof_get(_lib.K_KSTAR)[1] == 0
    rmu, rvar, rjac = orc.gp_predict(x[None], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
    rjv, rhm = orc.gp_linearize_extras(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    at = max(mu_atol(om), 1e-12)
    np.testing.assert_allclose(mu[:, 0], rmu[0], rtol=1e-9, atol=at)
    np.testing.assert_allclose(var[:, 0], rvar[0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(jm, rjac[0], rtol=1e-9, atol=10 * at)
    np.testing.assert_allclose(jv, rjv, rtol=1e-7, atol=1e-9 * float(np.max(syn["signal_var"])))
    np.testing.assert_allclose(hm, rhm, rtol=1e-8, atol=100 * at)
    np.testing.assert_array_equal(hm, np.swapaxes(hm, 1, 2))
    gp.set_small_path(2)
    out2 = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
    gp.set_small_path(1)
    for a_, b_, tol in zip((mu, var, jm, jv, hm), out2, (at, 1e-12, 10 * at, 1e-10, 100 * at)):
        np.testing.assert_allclose(a_, b_, rtol=1e-9, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("N,n_s,n_u", [(300, 2, 1), (384, 4, 1), (257, 3, 2)])
def test_single_query_against_384_rows_takes_the_streamed_route(N, n_s, n_u):
    """ONE query against a model padded to 384 rows (ARD-RBF, D <= 5): the streamed kernel (6 workgroups per output share
    the fetch of U^-1) instead of the one-launch pass (one workgroup per output); two or more queries keep the one-launch
    pass.  Both against the oracle and against each other."""
    from safe_exploration_amd import _lib
    syn = orc.make_synthetic(4000 + N, N, n_s, n_u, 8)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    x = np.hstack((syn["p"], syn["k_ff"]))
    at = max(mu_atol(om), 1e-12)
    for T, small in ((1, 0), (2, 1)):
        gp.prof_reset(); gp.prof_enable(True)
        mu, var, jac = gp.predict(x[:T], None, True)
        gp.prof_enable(False)
        assert gp.prof_get(_lib.K_SMALL)[1] == small and gp.prof_get(_lib.K_VAR)[1] == 1 - small
        rmu, rvar, rjac = orc.gp_predict(x[:T], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], True)
        np.testing.assert_allclose(mu, rmu, rtol=1e-9, atol=at)
        np.testing.assert_allclose(jac, rjac, rtol=1e-9, atol=10 * at)
        np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
        if T == 1:
            mu1, var1, jac1 = mu, var, jac
        else:
            np.testing.assert_allclose(mu[:1], mu1, rtol=1e-12, atol=1e-3 * at)
            np.testing.assert_allclose(jac[:1], jac1, rtol=1e-11, atol=1e-2 * at)
            np.testing.assert_allclose(var[:1], var1, rtol=0, atol=1e-12)
    # the one-step reachability of a single state goes the same way
    from safe_exploration_amd import gp_reachability as reach
    l = np.full(n_s, 0.05)
    q = 0.01 * np.eye(n_s)
    k_fb = 0.1 * np.ones((n_u, n_s))
    p1, q1 = reach.onestep_reachability(syn["p"][0][:, None], gp, syn["k_ff"][0][:, None], l, l, q_shape=q, k_fb=k_fb,
                                        c_safety=2.0, verbose=0)
    rp, rq, _ = orc.onestep_reachability_batch(om, syn["p"][:1], q[None], syn["k_ff"][:1], k_fb[None], l, l, 2.0)
    np.testing.assert_allclose(np.asarray(p1).ravel(), rp[0], rtol=1e-9, atol=at)
    np.testing.assert_allclose(np.asarray(q1), rq[0], rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("kt,N,T", [("mat52", 90, 5), ("lin_mat52", 150, 40), ("lin_rbf", 256, 300), ("mat52", 400, 200), ("mat52", 300, 17),
                                    ("lin_mat52", 1, 3)])
def test_fused_small_model_pass_general_kernels(kt, N, T):
    """the journal experiments' kernels through the one-launch pass (sr_gp_small_general_kernel): against the
    oracle's per-kernel formulas, finite-difference Jacobians, and the three-kernel route of the same library."""
    from safe_exploration_amd import SimpleGPModel, _lib
    rng = np.random.default_rng(900 + N)
    D = 3
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, 2))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(2)]
    noise = np.array([0.02, 0.03])
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * 2, hyp, noise + 1e-5)
    hh = [dict(h, noise_variance=nv) for h, nv in zip(hyp, noise)]
    gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hh)
    gp.train(Z, Y, opt_hyp=False)
    x = rng.uniform(-0.8, 0.8, (T, D))
    gp.prof_reset(); gp.prof_enable(True)
    mu, var, jac = gp.predict(x, None, True)
    gp.prof_enable(False)
    assert gp.prof_get(_lib.K_SMALL)[1] == 1 and gp.prof_get(_lib.K_VAR)[1] == 0
    rmu, rvar = orc.gp_predict_k(x, Z, beta, inv_K, [kt] * 2, hyp)
    scale = max(np.abs(beta).sum(0).max(), 1.0)
    np.testing.assert_allclose(mu, rmu, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-8 * max(1.0, float(rvar.max())))
    np.testing.assert_allclose(jac, orc.gp_mean_jacobian_k(x, Z, beta, [kt] * 2, hyp), rtol=1e-9, atol=1e-11 * scale)
    gp.set_small_path(2)
    mu2, var2, jac2 = gp.predict(x, None, True)
    gp.set_small_path(1)
    np.testing.assert_allclose(mu, mu2, rtol=1e-11, atol=1e-13 * scale)
    np.testing.assert_allclose(jac, jac2, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(var, var2, rtol=0, atol=1e-11 * max(1.0, float(rvar.max())))


@pytest.mark.gpu
def test_config3_cartpole_chain_full_model_size():
    """BASELINE configs[2]: cart-pole dims (n_s=4, n_u=1), N=5000, H=15.  A batch of 200 rollouts through
    sr_multistep_reach; three of them against the oracle chain (per-query route, factorised on the CPU), and the
    size-independent property that the one-call chain equals H one-step calls fed with their own outputs."""
    from safe_exploration_amd import gp_reachability as reach, workload
    N, T, H, n_s, n_u = 5000, 200, 15, 4, 1
    syn = orc.make_synthetic(3, N, n_s, n_u, 8, sf2=0.01)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    roll = workload.random_rollout_controls(33, T, H, n_s, n_u)
    l = np.full(n_s, 0.05)
    a, b = 0.5 * np.eye(n_s), np.zeros((n_s, n_u))
    p_all, q_all = reach.multistep_reachability_batch(roll["p0"], gp, roll["k_fb"], roll["k_ff"], l, l, None, 2.0, a, b)
    assert p_all.shape == (T, H, n_s) and q_all.shape == (T, H, n_s, n_s) and np.all(np.isfinite(q_all))
    assert np.linalg.eigvalsh(q_all.reshape(-1, n_s, n_s)).min() > 0
    # chain == repeated one-step calls (point branch first, then ellipsoid branch with k_fb[i-1])
    p, q = reach.onestep_reachability_batch(roll["p0"], gp, roll["k_ff"][:, 0], l, l, None, None, 2.0, a, b)
    np.testing.assert_allclose(p, p_all[:, 0], rtol=1e-13, atol=1e-15)
    for i in range(1, H):
        p, q = reach.onestep_reachability_batch(p, gp, roll["k_ff"][:, i], l, l, q, roll["k_fb"][:, i - 1], 2.0, a, b)
        np.testing.assert_allclose(p, p_all[:, i], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(q, q_all[:, i], rtol=1e-11, atol=1e-16)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    rp, rq = orc.multistep_reachability_batch(om, roll["p0"][:3], roll["k_fb"][:3], roll["k_ff"][:3], l, l, None, 2.0,
                                              a, b)
    np.testing.assert_allclose(p_all[:3], rp, rtol=1e-8, atol=max(mu_atol(om), 1e-11))
    np.testing.assert_allclose(q_all[:3], rq, rtol=1e-6, atol=1e-13)


@pytest.mark.gpu
def test_config4_identities_at_a_panelled_size():
    """BASELINE configs[3] uses identities of the exact posterior instead of an oracle (scripts/config4.py, N=50000);
    the same identities at N=3000 -- 24 row blocks = 6 panels of the two-level Cholesky, 5 levels of the recursive
    inversion -- where the oracle can still confirm them:  mu(z_i) + s2n alpha_i = y_i,  var(z_i) = s2n - s2n^2 (K_y^-1)_ii."""
    N = 3000
    syn = orc.make_synthetic(4, N, 2, 1, 4)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    s2n = syn["noise_var"] + 1e-8
    idx = np.random.default_rng(1).choice(N, 512, replace=False)
    mu, var = gp.predict(syn["Z"][idx])
    alpha = gp.beta
    assert np.abs(mu + s2n[None, :] * alpha[idx] - syn["Y"][idx]).max() < 1e-9
    _, wt = gp.export_state()
    off = gp._handle.Np - N
    for d in range(2):
        rows = __import__("torch").from_numpy(idx + off).to(wt.device)
        kinv_ii = (wt[d].index_select(0, rows) ** 2).sum(1).cpu().numpy()
        np.testing.assert_allclose(var[:, d], s2n[d] - s2n[d] ** 2 * kinv_ii, rtol=0, atol=1e-11)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    np.testing.assert_allclose(alpha, om["beta"], rtol=1e-6, atol=1e-8 * np.abs(om["beta"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kt,N,n_s,n_u", [("rbf", 700, 2, 1), ("rbf", 1300, 4, 1), ("rbf", 600, 3, 2), ("mat52", 650, 2, 1),
                                          ("lin_mat52", 900, 2, 1), ("lin_rbf", 130, 2, 1)])
def test_streamed_linearize_all_kernels(kt, N, n_s, n_u):
    """linearize_predict(jacobians=True) beyond the one-launch sizes: the columns [k*, dk*/dx_j] take ONE streaming pass
    over U^-1 (d var/dx_j = d k(x,x)/dx_j - 2 (U^-T dk*/dx_j).(U^-T k*)).  Against the oracle's closed forms and
    against the two-pass route of the same library (set_small_path(0))."""
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(70 + N)
    D = n_s + n_u
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, n_s))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(n_s)]
    noise = np.full(n_s, 0.02)
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * n_s, hyp, noise + 1e-5)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=[kt] * n_s, hyp=[dict(h, noise_variance=nv) for h, nv in zip(hyp, noise)])
    gp.train(Z, Y, opt_hyp=False)
    x = rng.uniform(-0.6, 0.6, D)
    mu, var, jm, jv, hm = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
    rmu, rvar = orc.gp_predict_k(x[None], Z, beta, inv_K, [kt] * n_s, hyp)
    rjv, rhm = orc.gp_linearize_extras_k(x, Z, beta, inv_K, [kt] * n_s, hyp)
    scale = max(np.abs(beta).sum(0).max(), 1.0)
    np.testing.assert_allclose(mu[:, 0], rmu[0], rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(var[:, 0], rvar[0], rtol=0, atol=1e-8 * max(1.0, float(rvar.max())))
    np.testing.assert_allclose(jm, orc.gp_mean_jacobian_k(x[None], Z, beta, [kt] * n_s, hyp)[0], rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(jv, rjv, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(rjv).max()))
    np.testing.assert_allclose(hm, rhm, rtol=1e-8, atol=1e-10 * scale)
    np.testing.assert_array_equal(hm, np.swapaxes(hm, 1, 2))
    gp.set_small_path(0)                                   # K1 -> K2 -> K3, U^-1 (U^-T k*), reduction kernel
    out0 = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
    gp.set_small_path(1)
    for a_, b_, tol in zip((mu, var, jm, jv, hm), out0, (1e-12 * scale, 1e-11, 1e-11 * scale, 1e-9, 1e-10 * scale)):
        np.testing.assert_allclose(a_, b_, rtol=1e-8, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("N,n_s,n_u", [(1300, 2, 1), (2600, 4, 1)])
def test_fused_stream_path_under_changing_inputs(N, n_s, n_u):
    """The fused small-batch route hands partial results from workgroup to workgroup inside one launch (tickets,
    agent-scope stores / loads, sr_stream.hip) and reuses the same scratch addresses call after call: a stale line
    from the PREVIOUS call would go unnoticed if every call evaluated the same query.  300 calls with different
    queries and batch sizes each, every result against the plain three-kernel MFMA path evaluated on the whole set at
    once; the single-query linearisation likewise against its own non-streamed route."""
    import torch
    rng = np.random.default_rng(N)
    D = n_s + n_u
    syn = orc.make_synthetic(N, N, n_s, n_u, 4, sf2=0.5)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    X = rng.uniform(-0.9, 0.9, (700, D))
    gp.set_small_path(0)
    rmu, rvar, rjac = gp.predict(X, None, True)
    gp.set_small_path(1)
    dev = gp.device
    tX = torch.from_numpy(X).to(dev)
    scale = max(float(np.abs(gp.beta).sum(0).max()), 1.0)
    pos, sizes = 0, [1, 3, 1, 9, 2, 40, 1, 4, 17, 1, 64, 5]
    k = 0
    while pos + 64 <= X.shape[0]:
        T = sizes[k % len(sizes)]
        k += 1
        mu, var, jac = gp.predict_device(tX[pos:pos + T], True)
        np.testing.assert_allclose(mu.cpu().numpy(), rmu[pos:pos + T], rtol=1e-10, atol=1e-12 * scale)
        np.testing.assert_allclose(var.cpu().numpy(), rvar[pos:pos + T], rtol=0, atol=2e-11)
        np.testing.assert_allclose(jac.cpu().numpy(), rjac[pos:pos + T], rtol=1e-10, atol=1e-11 * scale)
        pos += T
    assert k > 40
    # linearize: streamed route, a different query every call, interleaved with predict calls on the same handle
    refs = []
    gp.set_small_path(0)
    for q in range(12):
        refs.append(gp.linearize_predict(X[q:q + 1, :n_s], X[q:q + 1, n_s:], True))
    gp.set_small_path(1)
    for rep in range(3):
        for q in range(12):
            out = gp.linearize_predict(X[q:q + 1, :n_s], X[q:q + 1, n_s:], True)
            gp.predict_device(tX[100 + q:101 + q], True)
            for a_, b_, tol in zip(out, refs[q], (1e-12 * scale, 2e-11, 1e-11 * scale, 1e-9, 1e-10 * scale)):
                np.testing.assert_allclose(a_, b_, rtol=1e-8, atol=tol)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_s,n_xin,n_u", [("reach_tz_cart.npz", 4, 3, 1), ("reach_tz_n3.npz", 3, 2, 2)])
def test_gp_input_transform_vs_reference_golden(name, n_s, n_xin, n_u):
    """t_z_gp / a_gp_inp_x (gp_reachability_casadi.py:60-61,85,94-97; uncertainty_propagation_casadi.py:40-47,60): the GP
    sees t_z_gp @ state -- the journal cart-pole configuration drops the cart position (n_s = 4, D = 4).  Fixtures: the
    reference's numeric onestep / multistep functions on the wrapped model and its moment-propagation builders with
    a_gp_inp_x, both evaluated in the build container (tests/golden/make_golden.py, tz_case)."""
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, uncertainty_propagation_casadi as prop
    g = load_golden(name)
    tz = g["tz"]
    gp = SimpleGPModel(n_s, n_xin, n_u, kern_types=["rbf"] * n_s,
                       hyp=hyp_from(g["lengthscale"], g["signal_var"], g["noise_var"]))
    gp.train(g["Z"], g["Y"], opt_hyp=False)
    c = float(g["c_safety"])
    p1, q1 = reach.onestep_reachability_batch(g["p"], gp, g["k_ff"], g["l_mu"], g["l_sigma"], None, None, c,
                                              g["a_lin"], g["b_lin"], t_z_gp=tz)
    np.testing.assert_allclose(p1, g["p1_point"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(q1, g["q1_point"], rtol=1e-7, atol=1e-14)
    p1, q1 = reach.onestep_reachability_batch(g["p"], gp, g["k_ff"], g["l_mu"], g["l_sigma"], g["Q"], g["k_fb"], c,
                                              g["a_lin"], g["b_lin"], t_z_gp=tz)
    np.testing.assert_allclose(p1, g["p1_ell"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(q1, g["q1_ell"], rtol=1e-7, atol=1e-14)
    pa, qa = reach.multistep_reachability_batch(g["ms_p0"], gp, g["ms_k_fb"], g["ms_k_ff"], g["l_mu"], g["l_sigma"], None,
                                                c, g["a_lin"], g["b_lin"], t_z_gp=tz)
    np.testing.assert_allclose(pa, g["ms_p_all"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(qa, g["ms_q_all"], rtol=1e-6, atol=1e-13)
    # the transform is per call: the next call without it sees the identity again (and complains about the shapes)
    with pytest.raises((ValueError, RuntimeError)):
        reach.onestep_reachability_batch(g["p"], gp, g["k_ff"], g["l_mu"], g["l_sigma"], None, None, c)
    # single-query surface, HIP model and a foreign state-space model (GP on its own inputs, chain rule on the host)
    om = oracle_model(g["Z"], g["Y"], g["lengthscale"], g["signal_var"], g["noise_var"])

    def ssm(st, ac):
        m_, v_, j_ = orc._predict_one(om, np.hstack((st, ac))[0])
        return m_[:, None], v_[:, None], j_
    for model in (gp, ssm):
        pp, qq = reach.onestep_reachability(g["p"][1][:, None], model, g["k_ff"][1][:, None], g["l_mu"], g["l_sigma"],
                                            g["Q"][1], g["k_fb"][1], c, 0, g["a_lin"], g["b_lin"], tz)
        np.testing.assert_allclose(pp[:, 0], g["p1_ell"][1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(qq, g["q1_ell"][1], rtol=1e-7, atol=1e-14)
    # Gaussian moment propagation with a_gp_inp_x
    for tag, mode in (("taylor", prop.TAYLOR), ("meaneq", prop.MEAN_EQUIVALENT)):
        mu, sig, _ = prop.multistep_moments_batch(g["ms_p0"], gp, g["ms_k_ff"], g["ms_k_fb"], g["a_lin"], g["b_lin"], mode,
                                                  a_gp_inp_x=tz)
        np.testing.assert_allclose(mu, g["mu_" + tag], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(sig, g["sigma_" + tag], rtol=1e-7, atol=1e-13)
    H = g["ms_k_ff"].shape[1]
    mu_all, sigma_all, _ = prop.multi_step_taylor(g["ms_p0"][0][:, None], gp, g["ms_k_ff"][0], list(g["ms_k_fb"][0]), None,
                                                  g["a_lin"], g["b_lin"], tz)
    np.testing.assert_allclose(sigma_all.reshape(H, n_s, n_s), g["sigma_taylor"][0], rtol=1e-7, atol=1e-13)
    mu_f, sig_f, _ = prop.multi_step_taylor(g["ms_p0"][0][:, None], ssm, g["ms_k_ff"][0], list(g["ms_k_fb"][0]), None,
                                            g["a_lin"], g["b_lin"], tz)
    np.testing.assert_allclose(mu_f, g["mu_taylor"][0], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(sig_f.reshape(H, n_s, n_s), g["sigma_taylor"][0], rtol=1e-7, atol=1e-13)


# ------------------------------------------------------------------ persistent multi-step kernel (sr_chain.hip K0c)
@pytest.mark.parametrize("n_s,n_u,N,T,H,with_q0", [
    (2, 1, 100, 1, 2, False),        # Np = 128, one rollout
    (2, 1, 200, 256, 15, False),     # the regime of the reference's experiments
    (2, 1, 200, 17, 5, True),        # ragged last group, ellipsoid start
    (2, 1, 350, 300, 7, True),       # Np = 384
    (2, 1, 500, 416, 4, False),      # Np = 512: 26 groups x (2 outputs x 4 parts + tail) fill the launch
    (2, 1, 200, 1500, 3, False),     # 94 groups x (2 outputs x 2 parts + tail): two launches
    (4, 1, 150, 410, 6, True),       # cart-pole: 26 groups x (4 outputs x 2 parts + tail)
    (4, 1, 300, 100, 4, True),       # Np = 384 with n_s = 4
    (1, 1, 256, 481, 2, True),       # one output, two parts: exchange without a second output
    (1, 1, 200, 64, 3, False),
    (3, 1, 120, 40, 9, False),
    (1, 1, 90, 33, 5, True),         # one output: no hand-off between workgroups
    (2, 2, 180, 100, 5, True),
    (3, 2, 240, 64, 8, False),
])
def test_persistent_chain_matches_per_step_launches(n_s, n_u, N, T, H, with_q0):
    """(Tolerances: the per-step route of the larger cases takes other posterior kernels -- another order of
    summation, compounded over H steps; the persistent kernel also adds the shares of |U^-T k*|^2 of its Np / 128
    workgroups per output in its own order.)
    All H steps inside one launch (workgroups of a group of 16 rollouts hand their outputs round through L2)
    against the same chain launched step by step; and against the oracle's chain for a few rollouts."""
    from safe_exploration_amd import gp_reachability as reach
    syn = orc.make_synthetic(1000 + 7 * N + T, N, n_s, n_u, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    rng = np.random.default_rng(N + T + H)
    k_ff = 0.3 * rng.standard_normal((T, H, n_u))
    k_fb = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
    l_mu = np.linspace(0.005, 0.008, n_s)
    l_sg = np.linspace(0.002, 0.003, n_s)
    a = 0.6 * np.eye(n_s) + 0.03 * rng.standard_normal((n_s, n_s))      # contracting: the tubes stay bounded
    b = 0.1 * rng.standard_normal((n_s, n_u))
    q0 = syn["Q"] if with_q0 else None
    kfb0 = syn["k_fb"] if with_q0 else None
    args = (syn["p"], gp, k_fb, k_ff, l_mu, l_sg, q0, 2.0, a, b, kfb0)
    gp.set_chain(False)
    p_ref, q_ref = reach.multistep_reachability_batch(*args)
    assert not gp.last_chain
    gp.set_chain(True)
    chain_ok = True                  # every instantiation is scratch-free and dispatched (sr_chain_supported)
    for _ in range(3):               # tickets carry on from launch to launch
        p_all, q_all = reach.multistep_reachability_batch(*args)
        assert gp.last_chain == chain_ok
        assert np.all(np.isfinite(q_ref)) and np.all(np.isfinite(q_all))
        np.testing.assert_allclose(p_all, p_ref, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(q_all, q_ref, rtol=1e-7, atol=1e-14)
    # fewer rollouts after more (the tickets of the unused groups stay behind), then more again
    if T > 40 and T < 1024:
        sub = tuple(x[:20] if isinstance(x, np.ndarray) and x.shape[:1] == (T,) else x for x in args)
        p_s, q_s = reach.multistep_reachability_batch(*sub)
        np.testing.assert_allclose(p_s, p_ref[:20], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(q_s, q_ref[:20], rtol=1e-7, atol=1e-14)
        p_all, q_all = reach.multistep_reachability_batch(*args)
        np.testing.assert_allclose(q_all, q_ref, rtol=1e-7, atol=1e-14)
    # more rollouts than two launches hold (26 groups each here) while the per-step route still has its one-launch
    # posterior: the per-step route takes over (same results, checked above); two launches are still taken
    if T == 410:
        for reps, want_chain in ((2, True), (3, False)):
            big = tuple(np.concatenate([x] * reps)[:1000] if isinstance(x, np.ndarray) and x.shape[:1] == (T,) else x
                        for x in args)
            p_b, q_b = reach.multistep_reachability_batch(*big)
            assert gp.last_chain == want_chain
            np.testing.assert_allclose(p_b[:T], p_ref, rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(q_b[T:2 * T], q_ref, rtol=1e-7, atol=1e-14)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    n = min(T, 24)
    rp, rq = orc.multistep_reachability_batch(om, syn["p"][:n], k_fb[:n], k_ff[:n], l_mu, l_sg,
                                              None if q0 is None else q0[:n], 2.0, a, b,
                                              None if kfb0 is None else kfb0[:n])
    np.testing.assert_allclose(p_all[:n], rp, rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(q_all[:n], rq, rtol=1e-6, atol=1e-12)


def test_persistent_chain_timeout_is_reported_not_silent():
    """A persistent multi-step launch one of whose groups cannot complete its hand-offs (here: the launch is one workgroup
    short, sr_test_chain_drop -- in the field: the CUs were held by other work).  Within 100 ms the group gives up, fills
    its outputs with NaN AND raises the pinned status word: the NumPy entry point raises instead of returning NaN
    ellipsoids, sr_gp_chain_status reports it to device-tensor callers, the next entry point reports it to callers that
    never ask, the handle falls back to per-step launches, and the same call then succeeds."""
    import time
    import torch
    from safe_exploration_amd import gp_reachability as reach
    from safe_exploration_amd._lib import lib, check
    n_s, n_u, N, T, H = 2, 1, 200, 240, 6            # 15 groups x 2 outputs x 2 parts = 60 workgroups
    syn = orc.make_synthetic(4242, N, n_s, n_u, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    rng = np.random.default_rng(5)
    k_ff = 0.3 * rng.standard_normal((T, H, n_u))
    k_fb = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
    l = np.array([0.005, 0.008])
    a, b = 0.6 * np.eye(n_s), 0.1 * rng.standard_normal((n_s, n_u))
    args = (syn["p"], gp, k_fb, k_ff, l, l, None, 2.0, a, b)
    p_ref, q_ref = reach.multistep_reachability_batch(*args)
    assert gp.last_chain and not reach.chain_timed_out(gp._handle)
    check(lib.sr_test_chain_drop(gp._handle.h, 1))
    try:
        # (1) NumPy in / out: the call itself must raise
        t0 = time.time()
        with pytest.raises(RuntimeError, match="timed out"):
            reach.multistep_reachability_batch(*args)
        assert time.time() - t0 < 5.0                      # 100 ms, not 10 s
        # the handle now takes per-step launches: the same call works
        p2, q2 = reach.multistep_reachability_batch(*args)
        assert not gp.last_chain
        np.testing.assert_allclose(p2, p_ref, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(q2, q_ref, rtol=1e-7, atol=1e-14)
        # (2) device tensors in / out, nothing synchronises: NaN in the failed group only, status word set, and the
        # first entry after the failure reports it
        gp.set_chain(True)
        targs = tuple(torch.from_numpy(x).to(gp.device) if isinstance(x, np.ndarray) and x.shape[:1] == (T,) else x
                      for x in args)
        tp, tq = reach.multistep_reachability_batch(*targs)
        torch.cuda.synchronize()
        assert bool(torch.isnan(tq[-16:]).all()) and bool(torch.isfinite(tq[:-16]).all())
        np.testing.assert_allclose(tq[:-16].cpu().numpy(), q_ref[:-16], rtol=1e-7, atol=1e-14)
        with pytest.raises(RuntimeError, match="timed out"):
            reach.multistep_reachability_batch(*targs)
        tp, tq = reach.multistep_reachability_batch(*targs)       # per-step launches now
        torch.cuda.synchronize()
        assert not gp.last_chain
        np.testing.assert_allclose(tq.cpu().numpy(), q_ref, rtol=1e-7, atol=1e-14)
        # (3) a caller that asks: sr_gp_chain_status
        gp.set_chain(True)
        reach.multistep_reachability_batch(*targs)
        assert reach.chain_timed_out(gp._handle, gp.device)
        assert not reach.chain_timed_out(gp._handle, gp.device)          # cleared by the query
    finally:
        check(lib.sr_test_chain_drop(gp._handle.h, 0))
    # complete launches again: the persistent kernel can be re-armed, the groups resynchronise themselves
    gp.set_chain(True)
    for _ in range(2):
        p3, q3 = reach.multistep_reachability_batch(*args)
        assert gp.last_chain
        np.testing.assert_allclose(q3, q_ref, rtol=1e-7, atol=1e-14)


def test_persistent_chains_of_two_models_on_two_streams_take_turns():
    """Two handles, two streams, chains issued back to back without synchronising: each persistent launch needs (almost)
    every CU resident, so launches of one device wait for each other (an event gate in the library) instead of waiting
    for each other's workgroups until the time-out.  Every result must be right and nothing may time out."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, _buffers as B
    models, args, refs = [], [], []
    for seed, (n_s, N) in enumerate(((2, 200), (4, 150))):
        n_u, T, H = 1, 256, 12
        syn = orc.make_synthetic(900 + seed, N, n_s, n_u, T)
        gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
        rng = np.random.default_rng(seed)
        l = np.linspace(0.005, 0.008, n_s)
        a, b = 0.6 * np.eye(n_s), 0.1 * rng.standard_normal((n_s, n_u))
        dev = gp.device
        t_args = (B.as_dev(syn["p"], dev), gp, B.as_dev(0.1 * rng.standard_normal((T, H - 1, n_u, n_s)), dev),
                  B.as_dev(0.3 * rng.standard_normal((T, H, n_u)), dev), l, l, None, 2.0, a, b)
        models.append(gp)
        args.append(t_args)
        out = reach.multistep_reachability_batch(*t_args)
        torch.cuda.synchronize()
        assert gp.last_chain
        refs.append((out[0].clone(), out[1].clone()))
    streams = [torch.cuda.Stream(device=models[0].device) for _ in range(2)]
    outs = [[], []]
    for rep in range(20):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k].append(reach.multistep_reachability_batch(*args[k]))
    torch.cuda.synchronize()
    for k in range(2):
        assert not reach.chain_timed_out(models[k]._handle, models[k].device)
        for p_all, q_all in outs[k]:
            assert torch.equal(p_all, refs[k][0]) and torch.equal(q_all, refs[k][1])


def test_persistent_chain_beside_a_saturating_stream():
    """The chain kernel's workgroups wait for each other; while another stream keeps every CU busy (the 47 ms contraction
    of a 65536-query batch on an N = 5000 model) they may become resident late.  Whatever happens must be either the right
    answer or a reported failure followed by the right answer from the per-step route -- never silent NaN."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, workload
    big = orc.make_synthetic(5, 5000, 2, 1, 4)
    gbig = hip_model(big["Z"], big["Y"], big["lengthscale"], big["signal_var"], big["noise_var"], 2, 1)
    qb = workload.make_queries(9, 2, 1, 65536)
    xb = torch.from_numpy(np.hstack((qb["p"], qb["k_ff"]))).to(gbig.device)
    n_s, n_u, N, T, H = 2, 1, 200, 256, 15
    syn = orc.make_synthetic(777, N, n_s, n_u, T)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    rng = np.random.default_rng(6)
    k_ff = 0.3 * rng.standard_normal((T, H, n_u))
    k_fb = 0.1 * rng.standard_normal((T, H - 1, n_u, n_s))
    l = np.array([0.005, 0.008])
    a, b = 0.6 * np.eye(n_s), 0.1 * rng.standard_normal((n_s, n_u))
    args = (syn["p"], gp, k_fb, k_ff, l, l, None, 2.0, a, b)
    p_ref, q_ref = reach.multistep_reachability_batch(*args)
    assert gp.last_chain
    side = torch.cuda.Stream(device=gbig.device)
    failures = 0
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(3):
                gbig.predict_device(xb)                       # ~3 x 48 ms of MFMA work on every CU
        try:
            p_all, q_all = reach.multistep_reachability_batch(*args)
        except RuntimeError as e:
            assert "timed out" in str(e)
            failures += 1
            p_all, q_all = reach.multistep_reachability_batch(*args)      # per-step launches
            gp.set_chain(True)
        assert np.all(np.isfinite(q_all))
        np.testing.assert_allclose(p_all, p_ref, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(q_all, q_ref, rtol=1e-7, atol=1e-14)
        side.synchronize()
    print("chain launches that timed out beside the saturating stream: %d of 4" % failures)


def test_single_query_mailbox_and_fallback_agree():
    """__call__ / linearize_predict hand their results back through sr_publish + sr_wait_flag (pinned mailbox, host
    spin); the plain copy + stream sync route must give the same arrays, call after call.  (A model of 256 padded rows:
    every route runs the one-launch kernel, so the arrays are equal bit for bit; at 384 rows a single query through
    predict takes the streamed kernel -- test_single_query_against_384_rows_takes_the_streamed_route.)"""
    syn = orc.make_synthetic(17, 200, 2, 1, 8)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    io = gp._handle.single_io()
    assert io["mailbox"] and io["direct"]
    # (a) one command per call: query in the kernel arguments, results + sequence number written by the posterior kernel
    direct = [gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1]) for t in range(8)]
    lin_direct = gp.linearize_predict(syn["p"][:1], syn["k_ff"][:1], True)
    assert io["direct"] and io["seq"] == 9
    # (b) copy in, kernel, sr_publish: what a model without a one-launch posterior takes
    io["direct"] = False
    outs = []
    for t in range(8):
        outs.append(gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1]))
    assert io["mailbox"] and io["seq"] == 17           # the mailbox route stayed on
    lin = gp.linearize_predict(syn["p"][:1], syn["k_ff"][:1], True)
    for t in range(8):
        for a, b in zip(direct[t], outs[t]):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(lin_direct, lin):
        np.testing.assert_array_equal(a, b)
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu, var = gp.predict(x)
    for t in range(8):
        np.testing.assert_array_equal(outs[t][0][:, 0], mu[t])
        np.testing.assert_array_equal(outs[t][1][:, 0], var[t])
    io["mailbox"] = False
    for t in range(8):
        o = gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1])
        for a, b in zip(o, outs[t]):
            np.testing.assert_array_equal(a, b)
    lin2 = gp.linearize_predict(syn["p"][:1], syn["k_ff"][:1], True)
    for a, b in zip(lin, lin2):
        np.testing.assert_array_equal(a, b)
    io["mailbox"] = True
    # a model beyond the one-launch sizes: one command too -- the streamed kernels read the query from the pinned input block
    # and the workgroup that runs the final stage writes results and sequence number into the pinned result block
    syn2 = orc.make_synthetic(18, 900, 2, 1, 4)
    gp2 = hip_model(syn2["Z"], syn2["Y"], syn2["lengthscale"], syn2["signal_var"], syn2["noise_var"], 2, 1)
    io2 = gp2._handle.single_io()
    o2 = gp2(syn2["p"][:1], syn2["k_ff"][:1])
    l2 = gp2.linearize_predict(syn2["p"][1:2], syn2["k_ff"][1:2], True)
    assert io2["direct"] and io2["seq"] == 2
    io2["direct"] = False                                  # the copy + publish route of the same kernels
    o2b = gp2(syn2["p"][:1], syn2["k_ff"][:1])
    l2b = gp2.linearize_predict(syn2["p"][1:2], syn2["k_ff"][1:2], True)
    for a, b in zip(o2 + l2, o2b + l2b):
        np.testing.assert_array_equal(a, b)
    io2["direct"] = True
    mu2, var2 = gp2.predict(np.hstack((syn2["p"][:1], syn2["k_ff"][:1])))
    np.testing.assert_array_equal(o2[0][:, 0], mu2[0])
    np.testing.assert_array_equal(o2[1][:, 0], var2[0])
    om2 = oracle_model(syn2["Z"], syn2["Y"], syn2["lengthscale"], syn2["signal_var"], syn2["noise_var"])
    x2 = np.hstack((syn2["p"][1:2], syn2["k_ff"][1:2]))
    rjv, rhm = orc.gp_linearize_extras(x2[0], om2["Z"], om2["beta"], om2["inv_K"], om2["lengthscale"], om2["signal_var"])
    np.testing.assert_allclose(l2[3], rjv, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(rjv).max()))
    np.testing.assert_allclose(l2[4], rhm, rtol=1e-8, atol=1e3 * mu_atol(om2))
    # with the library's size dispatch switched off it declines, the refusal is remembered, the answer stays the same
    gp2.set_small_path(0)
    o2c = gp2(syn2["p"][:1], syn2["k_ff"][:1])
    assert not io2["direct"]
    np.testing.assert_allclose(o2c[0], o2[0], rtol=1e-10, atol=1e-12)
    gp2.set_small_path(1)
    # a flag that never comes is an error, not a hang
    from safe_exploration_amd._lib import lib
    from safe_exploration_amd import _buffers as B
    assert lib.sr_wait_flag(B.ptr(io["h_flag"]), io["seq"] + 12345, 0.05) != 0


@pytest.mark.parametrize("N,n_s,n_u", [(100, 2, 1), (150, 4, 1), (300, 2, 1), (450, 2, 1), (5, 2, 1), (128, 2, 1), (512, 2, 1),
                                      (200, 3, 2), (256, 2, 2)])
def test_resident_server_answers_single_queries(N, n_s, n_u):
    """K0s (sr_gp_server_start / sr_gp_server_call): the resident workgroups answer __call__, linearize_predict(jacobians=True)
    and a one-row predict from their mailbox.  Against the launched routes of the same model (to the last bits: the server
    always runs the second-order evaluation, on rows it fetched and scaled once) and against the oracle's closed forms."""
    import time
    import torch
    syn = orc.make_synthetic(40 + N, N, n_s, n_u, 12)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    om = oracle_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    D = n_s + n_u
    plain = [gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1]) for t in range(12)]
    plain_lin = [gp.linearize_predict(syn["p"][t:t + 1], syn["k_ff"][t:t + 1], True) for t in range(12)]
    assert gp.server_state()[:2] == (False, False)
    assert gp.start_server(idle_timeout_s=0.5) is True
    armed, resident, launches, calls = gp.server_state()
    assert armed and launches == 1 and calls == 0
    for rnd in range(3):
        for t in range(12):
            o = gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1])
            assert [a.shape for a in o] == [(n_s, 1), (n_s, 1), (n_s, D)]
            for a, b in zip(o, plain[t]):
                np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-3 * max(mu_atol(om), 1e-12))
            lin = gp.linearize_predict(syn["p"][t:t + 1], syn["k_ff"][t:t + 1], True)
            for a, b in zip(lin, plain_lin[t]):
                # (last bits: the server holds the training rows pre-scaled and, at 128 padded rows, its U^-1 fragments in
                #  registers with two accumulators per strip; at 512 the launched route is the streamed kernel)
                np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10)
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu1, var1, jac1 = gp.predict(x[3:4], compute_gradients=True)          # one row: the server again
    armed, resident, launches, calls = gp.server_state()
    assert armed and resident and launches == 1 and calls == 3 * 24 + 1
    rmu, rvar = orc.gp_predict(x[3:4], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
    np.testing.assert_allclose(mu1, rmu, rtol=1e-10, atol=mu_atol(om))
    np.testing.assert_allclose(var1, rvar, rtol=0, atol=1e-9 * float(np.max(om["signal_var"])))
    rjv, rhm = orc.gp_linearize_extras(x[5], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    lin5 = gp.linearize_predict(syn["p"][5:6], syn["k_ff"][5:6], True)
    np.testing.assert_allclose(lin5[3], rjv, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(rjv).max()))
    np.testing.assert_allclose(lin5[4], rhm, rtol=1e-8, atol=1e3 * mu_atol(om))
    # a batch goes the launched way beside the resident kernel
    mu_b, var_b = gp.predict(x)
    np.testing.assert_allclose(mu_b[3], mu1[0], rtol=1e-12, atol=1e-3 * max(mu_atol(om), 1e-12))
    # a device-wide wait elsewhere in the process does not hang on the resident kernel beyond its idle time-out
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 2.0
    gp.stop_server()
    assert gp.server_state()[:2] == (False, False)
    o = gp(syn["p"][:1], syn["k_ff"][:1])                                   # launched route again
    for a, b in zip(o, plain[0]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("kt,N,n_s,n_u", [("mat52", 100, 2, 1), ("lin_rbf", 128, 2, 1), ("lin_mat52", 150, 4, 1), ("lin_mat52", 25, 4, 1),
                                          ("lin_mat52", 150, 3, 1), ("mat52", 300, 2, 1), ("lin_rbf", 450, 2, 2), ("lin_mat52", 512, 2, 1),
                                          ("mat52", 5, 1, 1)])
def test_single_query_routes_with_the_journal_kernels(kt, N, n_s, n_u):
    """mat52 / lin_rbf / lin_mat52 -- the kernels of the reference's journal experiments
    (experiments/journal_experiment_configs/defaultconfig_episode.py:39, m = 150 inducing points, GP inputs = 3 transformed
    states + 1 action: the (150, 3, 1) row, D = 4; (150, 4, 1): D = 5) -- on the three fast single-query routes: the resident
    server, the one-command call (sr_gp_call1) and the one-launch second order (K0 LIN, general form).  Against the oracle's
    closed forms (gp_predict_k, gp_mean_jacobian_k, gp_linearize_extras_k), against the streamed multi-launch route of the
    same library (set_small_path(0)), and route against route."""
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(500 + N + n_s)
    D = n_s + n_u
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, n_s))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(n_s)]
    noise = np.full(n_s, 0.02)
    kts = [kt] * n_s
    beta, inv_K = orc.gp_fit_k(Z, Y, kts, hyp, noise + 1e-5)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=kts, hyp=[dict(h, noise_variance=nv) for h, nv in zip(hyp, noise)])
    gp.train(Z, Y, opt_hyp=False)
    X = rng.uniform(-0.7, 0.7, (6, D))
    X[5] = Z[N // 2]                                            # a query ON a training point (Matern: r = 0)
    scale = max(np.abs(beta).sum(0).max(), 1.0)

    def against_oracle(x, out, second):
        rmu, rvar = orc.gp_predict_k(x[None], Z, beta, inv_K, kts, hyp)
        np.testing.assert_allclose(out[0][:, 0], rmu[0], rtol=1e-9, atol=1e-11 * scale)
        np.testing.assert_allclose(out[1][:, 0], rvar[0], rtol=0, atol=1e-8 * max(1.0, float(rvar.max())))
        np.testing.assert_allclose(out[2], orc.gp_mean_jacobian_k(x[None], Z, beta, kts, hyp)[0], rtol=1e-9, atol=1e-10 * scale)
        if second:
            rjv, rhm = orc.gp_linearize_extras_k(x, Z, beta, inv_K, kts, hyp)
            np.testing.assert_allclose(out[3], rjv, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(rjv).max()))
            np.testing.assert_allclose(out[4], rhm, rtol=1e-8, atol=1e-10 * scale)
            np.testing.assert_array_equal(out[4], np.swapaxes(out[4], 1, 2))

    # 1. launched routes: one command (the default of a blocking call without a server) and the streamed reference route
    io = gp._handle.single_io()
    one_cmd = []
    for x in X:
        o1 = gp(x[None, :n_s], x[None, n_s:])
        o2 = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
        against_oracle(x, o1, False)
        against_oracle(x, o2, True)
        one_cmd.append((o1, o2))
    assert io["mailbox"], "the pinned blocks of this box are expected to be device-visible"
    if gp._handle.Np < 512:
        assert io["direct"], "sr_gp_call1 declined a %s model of %d padded rows" % (kt, gp._handle.Np)
    gp.set_small_path(0)
    ref = [(gp(x[None, :n_s], x[None, n_s:]), gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)) for x in X]
    gp.set_small_path(1)
    io["direct"] = True
    io.pop("direct_off", None)
    for (o1, o2), (r1, r2) in zip(one_cmd, ref):
        for a_, b_ in zip(o1 + o2, r1 + r2):
            np.testing.assert_allclose(a_, b_, rtol=1e-7, atol=1e-9 * scale)
    # 2. the same through sr_gp_linearize on device tensors (K0 LIN, general form, launched on the caller's stream)
    from safe_exploration_amd import _buffers as B
    mu_d, var_d, jm_d, jv_d, hm_d = gp.linearize_device(X[1])
    for a_, b_ in zip((B.to_numpy(jm_d), B.to_numpy(jv_d), B.to_numpy(hm_d)), one_cmd[1][1][2:]):
        np.testing.assert_allclose(a_, b_, rtol=1e-12, atol=1e-13 * scale)
    # 3. the resident server
    assert gp.start_server(idle_timeout_s=0.5) is True
    for rnd in range(2):
        for x, (o1, o2) in zip(X, one_cmd):
            s1 = gp(x[None, :n_s], x[None, n_s:])
            s2 = gp.linearize_predict(x[None, :n_s], x[None, n_s:], True)
            against_oracle(x, s1, False)
            against_oracle(x, s2, True)
            for a_, b_ in zip(s1 + s2, o1 + o2):
                np.testing.assert_allclose(a_, b_, rtol=1e-8, atol=1e-10 * scale)
    armed, resident, launches, calls = gp.server_state()
    assert armed and resident and launches == 1 and calls == 2 * 2 * len(X)
    # an input transform of the reachability entry points neither takes the server off the device nor changes its answers
    if n_u >= 1 and n_s >= 2:
        from safe_exploration_amd._lib import lib, check
        tz = B.as_dev(np.eye(n_s)[:n_s - 1], gp.device) if D - (n_s - 1) >= 1 and n_s - 1 >= 1 else None
        if tz is not None and n_s - 1 < D:
            check(lib.sr_gp_set_input_transform(gp._handle.h, B.ptr(tz), n_s - 1, B.stream_ptr(gp.device)))
            s1 = gp(X[0][None, :n_s], X[0][None, n_s:])
            check(lib.sr_gp_set_input_transform(gp._handle.h, None, 0, B.stream_ptr(gp.device)))
            for a_, b_ in zip(s1, one_cmd[0][0]):
                np.testing.assert_allclose(a_, b_, rtol=1e-8, atol=1e-10 * scale)
            assert gp.server_state()[1] and gp.server_state()[2] == 1
    gp.stop_server()


def test_model_on_a_device_that_is_not_current():
    """the NumPy routes synchronise the model's stream through its raw handle; handle 0 -- PyTorch's
    default stream -- is the null stream of the CURRENT device, so the wait must run with the model's device current.
    Needs two GPUs (skipped on the one-GPU box; the one-GPU part checks the handle carries its device)."""
    import torch
    from safe_exploration_amd import SimpleGPModel, _buffers as B
    rs = B.current_stream(torch.device("cuda:0"))
    assert rs.index == 0
    rs.synchronize()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    syn = orc.make_synthetic(5, 1500, 2, 1, 300)
    hyp = hyp_from(syn["lengthscale"], syn["signal_var"], syn["noise_var"])
    g1 = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=hyp, device="cuda:1")
    g1.train(syn["Z"], syn["Y"], opt_hyp=False)
    g0 = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    assert torch.cuda.current_device() == 0
    x = np.hstack((syn["p"], syn["k_ff"]))
    for _ in range(5):
        m1, v1 = g1.predict(x)
        m0, v0 = g0.predict(x)
        np.testing.assert_allclose(m1, m0, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(v1, v0, rtol=0, atol=1e-12)


def test_resident_server_idle_timeout_restart_and_model_updates():
    """The server leaves by itself after its idle time-out and comes back with the next query; a model update takes it off
    the device, and the next query is answered from the NEW model (row append inside the padded size, then a refit that
    changes the handle); a model that outgrows the one-launch sizes falls back to the launched routes for good."""
    import time
    syn = orc.make_synthetic(77, 120, 2, 1, 6)
    gp = hip_model(syn["Z"][:100], syn["Y"][:100], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    assert gp.start_server(idle_timeout_s=0.002)
    a0 = gp(syn["p"][:1], syn["k_ff"][:1])
    time.sleep(0.1)
    armed, resident, launches, calls = gp.server_state()
    assert armed and not resident and launches == 1 and calls == 1          # gone on its idle time-out
    a1 = gp(syn["p"][:1], syn["k_ff"][:1])
    for u, v in zip(a0, a1):
        np.testing.assert_array_equal(u, v)
    armed, resident, launches, calls = gp.server_state()
    assert armed and launches >= 2 and calls == 2
    # queries at the pace of the time-out: every one of them may find the kernel leaving
    for i in range(40):
        time.sleep(0.0021 if i % 2 else 0.0015)
        o = gp(syn["p"][:1], syn["k_ff"][:1])
        for u, v in zip(o, a0):
            np.testing.assert_array_equal(u, v)
    # row append: the model changes under the armed server
    gp.update_model(syn["Z"][100:101], syn["Y"][100:101], opt_hyp=False, replace_old=False)
    o = gp(syn["p"][:1], syn["k_ff"][:1])
    assert gp.server_state()[0]
    ref = hip_model(syn["Z"][:101], syn["Y"][:101], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    r = ref(syn["p"][:1], syn["k_ff"][:1])
    for u, v in zip(o, r):
        np.testing.assert_allclose(u, v, rtol=1e-9, atol=1e-11)
    # refit with another size: a new handle, the server follows it
    gp.train(syn["Z"][:120], syn["Y"][:120], opt_hyp=False)
    o = gp(syn["p"][1:2], syn["k_ff"][1:2])
    assert gp.server_state()[0] and gp.server_state()[3] >= 1
    ref = hip_model(syn["Z"][:120], syn["Y"][:120], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    r = ref(syn["p"][1:2], syn["k_ff"][1:2])
    for u, v in zip(o, r):
        np.testing.assert_allclose(u, v, rtol=1e-12, atol=1e-13)
    # a model beyond the one-launch sizes: no server, the launched routes answer
    big = orc.make_synthetic(78, 700, 2, 1, 2)
    gp.train(big["Z"], big["Y"], opt_hyp=False)
    assert not gp.server_state()[0]
    o = gp(big["p"][:1], big["k_ff"][:1])
    mu, var = gp.predict(np.hstack((big["p"][:1], big["k_ff"][:1])))
    np.testing.assert_array_equal(o[0][:, 0], mu[0])
    assert gp.start_server() is False
    # the journal experiments' kernels have a resident server too (round 5; the refusal this test used to assert is gone)
    from safe_exploration_amd import SimpleGPModel
    rng = np.random.default_rng(3)
    hyp = [dict(orc.make_hyp("lin_mat52", rng, 3), noise_variance=nv) for nv in (0.02, 0.03)]
    gm = SimpleGPModel(2, 2, 1, kern_types=["lin_mat52"] * 2, hyp=hyp)
    gm.train(syn["Z"][:60], syn["Y"][:60], opt_hyp=False)
    mu, var = gm.predict(np.hstack((syn["p"][:2], syn["k_ff"][:2])))          # launched, two rows
    assert gm.start_server() is True and gm.server_state()[0] is True
    o = gm(syn["p"][:1], syn["k_ff"][:1])
    assert gm.server_state()[3] == 1
    np.testing.assert_allclose(o[0][:, 0], mu[0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(o[1][:, 0], var[0], rtol=0, atol=1e-11)
    gm.stop_server()


def test_resident_server_request_given_up_and_callers_on_two_threads():
    """A request that is given up (time-out) never lends its answers to the next one: the library retires the sequence
    number and calls the launch off; the next call starts a fresh launch and answers ITS query.  Two threads calling the
    same model take turns (one mailbox)."""
    import ctypes
    import threading
    import time
    from safe_exploration_amd import _lib
    syn = orc.make_synthetic(79, 100, 2, 1, 8)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    ref = [gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1]) for t in range(8)]        # launched route
    assert gp.start_server(idle_timeout_s=0.002)
    hd = gp._handle
    io = hd.single_io()
    given_up = 0
    for rep in range(20):
        time.sleep(0.01)                                    # the server has left: the next request needs a launch
        x = np.concatenate((syn["p"][rep % 8], syn["k_ff"][rep % 8]))
        io["h_in_np"][:x.size] = x
        # a negative time-out gives up at the first look at the clock (about 1000 polls): sometimes before the fresh
        # launch answers, sometimes after -- both must leave the protocol intact
        rc = _lib.lib.sr_gp_server_call(hd.h, io["p_in"], 1, io["p_srv"], ctypes.c_double(-1.0))
        assert rc in (0, _lib.SR_ESTATE), rc
        given_up += rc != 0
        t = (rep + 3) % 8
        o = gp(syn["p"][t:t + 1], syn["k_ff"][t:t + 1])
        for u, v in zip(o, ref[t]):
            np.testing.assert_allclose(u, v, rtol=1e-12, atol=1e-14)
    assert gp.server_state()[0]
    print("requests given up: %d of 20" % given_up)

    errors = []

    def worker(k):                                          # the C-ABI with buffers of the thread's own
        try:
            x, out = np.empty(3), np.empty(2 * 2 + 2 * 2 * 3 + 2 * 9)
            px, po = ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(out.ctypes.data)
            for i in range(300):
                t = (i + k) % 8
                x[:2], x[2:] = syn["p"][t], syn["k_ff"][t]
                rc = _lib.lib.sr_gp_server_call(hd.h, px, 0, po, ctypes.c_double(5.0))
                assert rc == 0, rc
                np.testing.assert_allclose(out[:2], np.ravel(ref[t][0]), rtol=1e-12, atol=1e-14)
                np.testing.assert_allclose(out[2:4], np.ravel(ref[t][1]), rtol=1e-12, atol=1e-14)
        except Exception as e:                              # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[0]


def test_resident_servers_of_two_models_and_a_deep_copy():
    """Two models keep a resident server each (different sizes, different numbers of outputs); a deep copy of a model --
    what CasadiSSMEvaluator holds (state_space_models.py:166) -- shares the handle and with it the server; destroying a
    model whose server is resident takes the kernel off the device (no hang, no leak of the pinned block)."""
    import copy
    import gc
    import torch
    s1 = orc.make_synthetic(91, 90, 2, 1, 6)
    s2 = orc.make_synthetic(92, 140, 4, 1, 6)
    g1 = hip_model(s1["Z"], s1["Y"], s1["lengthscale"], s1["signal_var"], s1["noise_var"], 2, 1)
    g2 = hip_model(s2["Z"], s2["Y"], s2["lengthscale"], s2["signal_var"], s2["noise_var"], 4, 1)
    r1 = [g1(s1["p"][t:t + 1], s1["k_ff"][t:t + 1]) for t in range(6)]
    r2 = [g2.linearize_predict(s2["p"][t:t + 1], s2["k_ff"][t:t + 1], True) for t in range(6)]
    assert g1.start_server(0.2) and g2.start_server(0.2)
    c1 = copy.deepcopy(g1)
    assert c1._handle is g1._handle and c1.server_state()[0]
    for rnd in range(3):
        for t in range(6):
            for m in (g1, c1):
                o = m(s1["p"][t:t + 1], s1["k_ff"][t:t + 1])
                for a, b in zip(o, r1[t]):
                    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
            o = g2.linearize_predict(s2["p"][t:t + 1], s2["k_ff"][t:t + 1], True)
            for a, b in zip(o, r2[t]):
                np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-10)
    assert g1.server_state()[1] and g2.server_state()[1]            # both resident
    assert g1.server_state()[3] == 36 and g2.server_state()[3] == 18
    # a persistent multi-step launch needs (almost) every CU: the resident servers leave for it and come back afterwards
    from safe_exploration_amd import gp_reachability as reach, workload
    roll = workload.random_rollout_controls(5, 64, 8, 2, 1)
    l = np.array([0.05, 0.02])
    pa, qa = reach.multistep_reachability_batch(roll["p0"], g1, roll["k_fb"], roll["k_ff"], l, l, None, 2.0, 0.8 * np.eye(2), np.zeros((2, 1)))
    assert np.all(np.isfinite(qa)) and g1._handle and not g1.server_state()[1] and g1.server_state()[0]
    o = g1(s1["p"][:1], s1["k_ff"][:1])
    assert g1.server_state()[1]
    for a, b in zip(o, r1[0]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
    del g2
    gc.collect()
    torch.cuda.synchronize()                                         # g1's server leaves on its idle time-out at the latest
    o = g1(s1["p"][:1], s1["k_ff"][:1])
    for a, b in zip(o, r1[0]):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)
    del c1
    g1.stop_server()
