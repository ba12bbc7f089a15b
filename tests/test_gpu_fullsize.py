"""Full-size configurations of BASELINE.json under the driver's GPU suite, and the CasADi callback boundary on
the HIP model.

  * C5 per-GPU share at the HEADLINE model: N = 5000, 1 048 576 query states through 16 chunks of the bounded
    workspace (oracle sample + chunk-boundary identities),
  * C4: N = 50000 training points, the model update itself (posterior identities; no CPU oracle reaches it),
  * two-rank RCCL: replicate_model over backend "nccl" + sharded one-step == single process (needs 2 GPUs),
  * CasadiSSMEvaluator (forward / Jacobian / reverse callbacks) on the HIP SimpleGPModel against the oracle and
    finite differences.
"""
import os
import socket
import sys

import numpy as np
import pytest

from _helpers import hip_model, mu_atol, cached_oracle_model, hyp_from, oracle_model
from conftest import ROOT
from oracle import oracle_np as orc

pytestmark = pytest.mark.gpu
STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standin")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(lib_built):
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"


def test_config5_share_at_the_headline_model():
    """BASELINE configs[4], one GPU's share: N = 5000 (the headline model), T = 1 048 576 query states streamed
    through 16 chunks of 65536.  (i) 4096 random rows against the oracle (explicit-inverse route, factorised on the
    CPU) at the SURVEY 8(d) tolerances, (ii) rows of two chunks (one interior, the last) bit-equal to a direct call
    on exactly those rows, (iii) a window straddling a chunk boundary against a direct call on the window,
    (iv) positivity of every variance and of every shape matrix."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, workload
    N, T, C = 5000, 1 << 20, 65536
    syn = orc.make_synthetic(5, N, 2, 1, 4)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    q = workload.make_queries(555, 2, 1, T)
    dev = gp.device
    tp, tq, tkff, tkfb = (torch.from_numpy(q[k]).to(dev) for k in ("p", "Q", "k_ff", "k_fb"))
    l = np.array([0.05, 0.02])
    p1, q1, var = reach.onestep_reachability_batch(tp, gp, tkff, l, l, tq, tkfb, 2.0, return_var=True)
    assert p1.shape == (T, 2) and q1.shape == (T, 2, 2) and var.shape == (T, 2)
    assert bool(torch.isfinite(q1).all()) and float(var.min()) > 0 and float(var.max()) <= 1.0 + 1e-12
    det = q1[:, 0, 0] * q1[:, 1, 1] - q1[:, 0, 1] * q1[:, 1, 0]
    assert float(det.min()) > 0 and float(q1[:, 0, 0].min()) > 0
    assert float((q1[:, 0, 1] - q1[:, 1, 0]).abs().max()) <= 1e-15 * float(q1.abs().max())
    # (ii) chunk-aligned slices: same kernels, same tile positions -> bit-equal
    for c in (7, 15):
        sl = slice(c * C, (c + 1) * C)
        dp, dq, dv = reach.onestep_reachability_batch(tp[sl], gp, tkff[sl], l, l, tq[sl], tkfb[sl], 2.0, return_var=True)
        assert torch.equal(dp, p1[sl]) and torch.equal(dq, q1[sl]) and torch.equal(dv, var[sl])
    # (iii) a window across the boundary between chunks 8 and 9 (rows sit at other tile positions)
    sl = slice(9 * C - C // 2, 9 * C + C // 2)
    dp, dq, dv = reach.onestep_reachability_batch(tp[sl], gp, tkff[sl], l, l, tq[sl], tkfb[sl], 2.0, return_var=True)
    np.testing.assert_allclose(dp.cpu().numpy(), p1[sl].cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(dv.cpu().numpy(), var[sl].cpu().numpy(), rtol=0, atol=1e-13)
    np.testing.assert_allclose(dq.cpu().numpy(), q1[sl].cpu().numpy(), rtol=1e-10, atol=1e-13)
    # (i) oracle sample, rows drawn from every chunk
    om = cached_oracle_model(5, N, 2, 1)
    idx = np.sort(np.random.default_rng(1).choice(T, 4096, replace=False))
    assert len(set(idx // C)) == 16
    rp, rq, rvar = orc.onestep_reachability_vectorised(om, q["p"][idx], q["Q"][idx], q["k_ff"][idx], q["k_fb"][idx],
                                                       l, l, 2.0, np.eye(2), np.zeros((2, 1)))
    ti = torch.from_numpy(idx).to(dev)
    np.testing.assert_allclose(p1[ti].cpu().numpy(), rp, rtol=1e-9, atol=max(mu_atol(om), 1e-12))
    np.testing.assert_allclose(var[ti].cpu().numpy(), rvar, rtol=0, atol=1e-9)      # 1e-9 * sigma_f^2
    np.testing.assert_allclose(q1[ti].cpu().numpy(), rq, rtol=1e-8, atol=30 * 1e-9)  # see test_full_size_headline_config


def test_config4_model_update_at_50000_training_points():
    """BASELINE configs[3] itself: N = 50000, n_out = 2 -- Gram matrix, blocked fp64 Cholesky with MFMA trailing
    updates, explicit U^-1 (2 x 20 GB resident).  No CPU oracle reaches this size; the exact posterior has
    size-independent identities at the training inputs (K_y alpha = y and the diagonal of K_y^-1):
        mu(z_i) + s2n alpha_i = y_i ,    var(z_i) = s2n - s2n^2 (K_y^-1)_ii ,  (K_y^-1)_ii = |row i of U^-1|^2
    plus: U^-1 is upper triangular with positive diagonal, and K_y (U^-1 e_j) reproduces U^-T e_j on sampled columns
    (i.e. U^-1 U^-T = K_y^-1 against the Gram matrix rebuilt on the host for 64 columns)."""
    import torch
    from safe_exploration_amd import workload, SimpleGPModel
    N, n_s, n_u = 50000, 2, 1
    prob = workload.make_problem(4, N, n_s, n_u, 4)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    hd = gp._handle
    Np, off = hd.Np, hd.Np - N
    # the update keeps its scratch (U, W: 2 x 20 GB) for the next refit; a host that only evaluates hands it back
    free0 = torch.cuda.mem_get_info(gp.device)[0]
    gp.release_scratch()
    assert torch.cuda.mem_get_info(gp.device)[0] - free0 > 35e9
    s2n = prob["noise_var"] + 1e-5 + 1e-8
    idx = np.random.default_rng(0).choice(N, 2048, replace=False)
    mu, var = gp.predict(prob["Z"][idx])
    alpha = gp.beta
    assert np.abs(mu + s2n[None, :] * alpha[idx] - prob["Y"][idx]).max() < 1e-9
    _, wt = gp.export_state()
    rows = torch.from_numpy(idx + off).to(wt.device)
    for d in range(n_s):
        kinv_ii = (wt[d].index_select(0, rows) ** 2).sum(1).cpu().numpy()
        assert np.abs(var[:, d] - (s2n[d] - s2n[d] ** 2 * kinv_ii)).max() < 1e-9
        diag = torch.diagonal(wt[d])
        assert float(diag.min()) > 0
        # strictly-lower part is exactly zero on sampled rows
        r = int(rows[0])
        assert float(wt[d][r, :r].abs().max()) == 0.0
    # K_y^-1 y = alpha through the factor itself: alpha = U^-1 (U^-T y) on the device, one output
    y0 = torch.zeros(Np, dtype=torch.float64, device=wt.device)
    y0[off:] = torch.from_numpy(np.ascontiguousarray(prob["Y"][:, 0])).to(wt.device)
    a0 = torch.mv(wt[0], torch.mv(wt[0].T, y0))[off:].cpu().numpy()
    np.testing.assert_allclose(a0, alpha[:, 0], rtol=1e-9, atol=1e-9 * np.abs(alpha[:, 0]).max())
    # residual of the linear system against the Gram matrix rebuilt on the host (64 rows of K_y)
    rs = np.random.default_rng(2).choice(N, 64, replace=False)
    Zs = prob["Z"] / prob["lengthscale"][0][None, :]
    d2 = ((Zs[rs][:, None, :] - Zs[None, :, :]) ** 2).sum(-1)
    Krows = prob["signal_var"][0] * np.exp(-0.5 * d2)
    Krows[np.arange(64), rs] += s2n[0]
    res = Krows.dot(alpha[:, 0]) - prob["Y"][rs, 0]
    assert np.abs(res).max() < 1e-8 * max(1.0, np.abs(alpha[:, 0]).max())
    del wt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _nccl_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, parallel
        n_s, n_u, N, T = 2, 1, 1000, 5001
        prob = workload.make_problem(31, N, n_s, n_u, T, sf2=0.01)
        gp = None
        if rank == 0:
            gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device=dev)
            gp.train(prob["Z"], prob["Y"], opt_hyp=False, noise_diag=2e-5)
        gp = parallel.replicate_model(gp, src=0)           # RCCL broadcast of Z / targets / alpha / U^-1
        assert gp.device == dev and gp._noise_diag == 2e-5
        lo, hi = parallel.shard_bounds(T, world, rank)
        l = np.array([0.05, 0.02])
        p1, q1 = reach.onestep_reachability_batch(prob["p"][lo:hi], gp, prob["k_ff"][lo:hi], l, l,
                                                  prob["Q"][lo:hi], prob["k_fb"][lo:hi], 2.0)
        full_p = parallel.gather_rows(p1, dst=0)
        full_q = parallel.gather_rows(q1, dst=0)
        if rank == 0:
            rp, rq = reach.onestep_reachability_batch(prob["p"], gp, prob["k_ff"], l, l, prob["Q"], prob["k_fb"], 2.0)
            ret["dp"] = float(np.abs(full_p - rp).max())
            ret["dq"] = float(np.abs(full_q - rq).max() / np.abs(rq).max())
            ret["backend"] = dist.get_backend()
            ret["world"] = dist.get_world_size()
        else:
            # the receiver holds a complete model: a later append works on consistent targets / noise
            gp.update_model(prob["Z"][:3] + 0.01, prob["Y"][:3], opt_hyp=False, replace_old=False, noise_diag=2e-5)
            assert gp.beta.shape == (N + 3, n_s)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ranks_over_rccl():
    """One rank per visible GPU (at most 8) over backend "nccl" (RCCL): rank 0 factorises on cuda:0, the others adopt
    the model through the packed-triangle broadcast on their own device; the sharded one-step result equals the
    single-process one.  Skipped on a one-GPU box (RCCL refuses two ranks on one device; the same code path runs
    there over gloo: tests/test_gpu_distributed.py)."""
    import torch
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_nccl_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert ret["backend"] == "nccl" and ret["world"] == world
        assert ret["dp"] < 1e-13 and ret["dq"] < 1e-12


@pytest.mark.parametrize("kt,N,n_s,n_u", [("rbf", 200, 2, 1), ("rbf", 900, 4, 1), ("mat52", 700, 2, 1), ("lin_mat52", 150, 2, 1)])
def test_casadi_evaluator_callbacks_on_the_hip_model(kt, N, n_s, n_u, monkeypatch):
    """The boundary the MPC's IPOPT loop goes through (state_space_models.py:214-566): the forward, Jacobian and
    reverse callbacks of ``get_forward_model_casadi`` evaluated ON the HIP model (a stand-in module supplies the casadi
    base classes; one model evaluation per callback).  The stacked (2n + nD) x D Jacobian is compared with the
    oracle's closed forms and with central differences of the forward callback; the reverse callback with seed^T J."""
    from safe_exploration_amd import SimpleGPModel
    monkeypatch.syspath_prepend(STANDIN)
    monkeypatch.delitem(sys.modules, "casadi", raising=False)
    import casadi  # noqa: F401  (the stand-in must be the module the evaluator finds)
    rng = np.random.default_rng(N + n_s)
    D = n_s + n_u
    Z = rng.uniform(-1, 1, (N, D))
    Y = rng.standard_normal((N, n_s))
    hyp = [orc.make_hyp(kt, rng, D) for _ in range(n_s)]
    noise = np.full(n_s, 0.02)
    beta, inv_K = orc.gp_fit_k(Z, Y, [kt] * n_s, hyp, noise + 1e-5)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=[kt] * n_s, hyp=[dict(h, noise_variance=nv) for h, nv in zip(hyp, noise)])
    gp.train(Z, Y, opt_hyp=False)
    ev = gp.get_forward_model_casadi(True)
    assert ev.ssm is not gp and ev.ssm._handle is gp._handle and ev.v_has_reverse and ev.v_has_jacobian
    z = rng.uniform(-0.6, 0.6, D)
    x, u = z[:n_s, None], z[n_s:, None]
    mu, var, jac = (np.array(o) for o in ev(x, u))                   # shape-checked numeric call
    scale = max(np.abs(beta).sum(0).max(), 1.0)
    rmu, rvar = orc.gp_predict_k(z[None], Z, beta, inv_K, [kt] * n_s, hyp)
    rjm = orc.gp_mean_jacobian_k(z[None], Z, beta, [kt] * n_s, hyp)[0]
    rjv, rhm = orc.gp_linearize_extras_k(z, Z, beta, inv_K, [kt] * n_s, hyp)
    np.testing.assert_allclose(mu[:, 0], rmu[0], rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(var[:, 0], rvar[0], rtol=0, atol=1e-8 * max(1.0, float(rvar.max())))
    np.testing.assert_allclose(jac, rjm, rtol=1e-9, atol=1e-11 * scale)
    assert ev.jac_mu_order == "F"                 # CasADi's vec rule is the default; "C" = the reference helper's rows
    ev.jac_mu_order = "C"
    jfun = ev.get_jacobian("jac_CasadiModelEvaluator", [], [], {})
    (stacked,) = (np.array(o) for o in jfun(x, u, mu, var, jac))
    assert stacked.shape == (2 * n_s + n_s * D, D)
    want = np.vstack((rjm, rjv, rhm.reshape(n_s * D, D)))
    np.testing.assert_allclose(stacked[:n_s], want[:n_s], rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(stacked[n_s:2 * n_s], want[n_s:2 * n_s], rtol=1e-6, atol=1e-8 * max(1.0, np.abs(rjv).max()))
    np.testing.assert_allclose(stacked[2 * n_s:], want[2 * n_s:], rtol=1e-8, atol=1e-10 * scale)
    # central differences of the forward callback itself: d [mu; var; vec(jac_mu)] / dz
    eps = 1e-5
    fd = np.empty_like(stacked)
    for j in range(D):
        e = np.zeros(D)
        e[j] = eps
        op = [np.array(o) for o in ev.eval([(z + e)[:n_s, None], (z + e)[n_s:, None]])]
        om = [np.array(o) for o in ev.eval([(z - e)[:n_s, None], (z - e)[n_s:, None]])]
        fd[:, j] = np.concatenate([(a - b).reshape(-1) for a, b in zip(op, om)]) / (2 * eps)
    np.testing.assert_allclose(stacked, fd, rtol=2e-5, atol=2e-6 * max(scale, np.abs(stacked).max()))
    # reverse-mode callback == seed^T J
    bfun = ev.get_reverse(1, "adj1_CasadiModelEvaluator", [], [], {})
    s_mu, s_var, s_jac = rng.standard_normal((n_s, 1)), rng.standard_normal((n_s, 1)), rng.standard_normal((n_s, D))
    adj_x, adj_u = (np.array(o) for o in bfun(x, u, mu, var, jac, s_mu, s_var, s_jac))
    g = np.concatenate((s_mu.ravel(), s_var.ravel(), s_jac.ravel())).dot(stacked)
    np.testing.assert_allclose(adj_x[:, 0], g[:n_s], rtol=1e-12, atol=1e-13 * np.abs(g).max())
    np.testing.assert_allclose(adj_u[:, 0], g[n_s:], rtol=1e-12, atol=1e-13 * np.abs(g).max())
    # column-major numbering of the jac_mean block (CasADi's own vec of a dense output): a row permutation only
    ev.jac_mu_order = "F"
    (stacked_f,) = (np.array(o) for o in jfun(x, u, mu, var, jac))
    ev.jac_mu_order = "C"
    perm = np.arange(n_s * D).reshape(n_s, D).T.reshape(-1)
    np.testing.assert_array_equal(stacked_f[:2 * n_s], stacked[:2 * n_s])
    np.testing.assert_array_equal(stacked_f[2 * n_s:], stacked[2 * n_s:][perm])
    # not linearised: two outputs, (2n) x D Jacobian from predict(states, actions, jacobians=True)
    ev2 = gp.get_forward_model_casadi(False)
    m2, v2 = (np.array(o) for o in ev2(x, u))
    np.testing.assert_allclose(m2, mu, rtol=1e-12, atol=1e-13 * scale)
    (st2,) = (np.array(o) for o in ev2.get_jacobian("jac2", [], [], {})(x, u, m2, v2))
    np.testing.assert_allclose(st2, stacked[:2 * n_s], rtol=1e-12, atol=1e-13 * scale)
    ax2, au2 = (np.array(o) for o in ev2.get_reverse(1, "adj2", [], [], {})(x, u, m2, v2, s_mu, s_var))
    g2 = np.concatenate((s_mu.ravel(), s_var.ravel())).dot(st2)
    np.testing.assert_allclose(np.concatenate((ax2.ravel(), au2.ravel())), g2, rtol=1e-12, atol=1e-13 * np.abs(g2).max())


def test_packed_export_import_round_trip():
    """The replication format (sr_gp_export_packed / sr_gp_import_begin / _packed / _end): pieces of the packed upper
    triangle into a fresh handle, in shuffled order, must reproduce the sender's dense U^-1 and alpha bit for bit --
    and with them every prediction.  N not a multiple of 128 (front padding in play)."""
    import torch
    from safe_exploration_amd import SimpleGPModel, parallel
    for N, n_s, n_u in ((333, 2, 1), (1500, 3, 2)):
        syn = orc.make_synthetic(40 + N, N, n_s, n_u, 300)
        gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
        alpha, wt = gp.export_state()
        pieces = parallel.packed_pieces(N, 15000)
        assert len(pieces) >= 3
        assert gp.packed_count(0, N) == N * (N + 1) // 2
        other = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s,
                              hyp=hyp_from(syn["lengthscale"], syn["signal_var"], syn["noise_var"]))
        other.begin_import(syn["Z"], syn["Y"], gp.export_alpha())
        jobs = [(d, r0, r1, c) for d in range(n_s) for (r0, r1, c) in pieces]
        np.random.default_rng(0).shuffle(jobs)
        for d, r0, r1, c in jobs:
            buf = torch.empty(c, dtype=torch.float64, device=gp.device)
            gp.export_packed(d, r0, r1, buf)
            # the packed piece is exactly rows r0..r1-1 of the upper triangle of the real block
            off = wt.shape[1] - N
            want = torch.cat([wt[d, off + i, off + i:] for i in range(r0, min(r1, r0 + 3))])
            assert torch.equal(buf[:want.numel()], want)
            other.import_packed(d, r0, r1, buf)
        other.end_import()
        a2, w2 = other.export_state()
        assert torch.equal(a2, alpha) and torch.equal(w2, wt)
        x = np.hstack((syn["p"], syn["k_ff"]))
        m1, v1 = gp.predict(x)
        m2, v2 = other.predict(x)
        assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    with pytest.raises(ValueError):
        gp.packed_count(5, 3)


def test_first_small_batch_reach_call_on_a_big_model():
    """on a model with 2 ceil(Np / 256) > pick_nsplit (n_out = 8: Np > 12288) the fused T <= 64 route needs
    more mean partials than the reachability entry points used to size the workspace for -- the first such call on a
    fresh handle reallocated the workspace under the pointers its caller held.  First call on the handle: T = 8
    one-step reachability; then the same rows through the plain three-kernel route and a 3-step chain."""
    import torch
    from safe_exploration_amd import gp_reachability as reach
    n_s, n_u, N, T = 8, 1, 12400, 8
    syn = orc.make_synthetic(77, N, n_s, n_u, T, sf2=0.01)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    l = np.full(n_s, 0.05)
    a, b = 0.5 * np.eye(n_s), np.zeros((n_s, n_u))
    p1, q1, v1 = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0, a, b,
                                                  return_var=True)
    assert np.isfinite(q1).all() and (v1 > 0).all()
    gp.set_small_path(0)
    p0, q0, v0 = reach.onestep_reachability_batch(syn["p"], gp, syn["k_ff"], l, l, syn["Q"], syn["k_fb"], 2.0, a, b,
                                                  return_var=True)
    gp.set_small_path(1)
    np.testing.assert_allclose(p1, p0, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(v1, v0, rtol=0, atol=1e-11)
    np.testing.assert_allclose(q1, q0, rtol=1e-8, atol=1e-14)
    # the posterior itself against predict() on the same inputs (a different entry point, its own buffers)
    mu, var = gp.predict(np.hstack((syn["p"], syn["k_ff"])))
    np.testing.assert_allclose(v1, var, rtol=0, atol=1e-11)
    # a short chain as the first multi-step call (T <= 64 per step)
    gp2 = hip_model(syn["Z"][:12300], syn["Y"][:12300], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    rng = np.random.default_rng(3)
    k_ff = 0.1 * rng.standard_normal((T, 3, n_u))
    k_fb = 0.1 * rng.standard_normal((T, 2, n_u, n_s))
    pa, qa = reach.multistep_reachability_batch(syn["p"], gp2, k_fb, k_ff, l, l, None, 2.0, a, b)
    gp2.set_small_path(0)
    pb, qb = reach.multistep_reachability_batch(syn["p"], gp2, k_fb, k_ff, l, l, None, 2.0, a, b)
    np.testing.assert_allclose(pa, pb, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(qa, qb, rtol=1e-7, atol=1e-13)


def test_streamed_small_batches_at_the_headline_model_size():
    """5 .. 64 queries against the N = 5000 model take the streamed route with RUNS of k-chunks per workgroup
    (sr_stream_mfma_kernel, only dispatched from N ~ 4000 on) and the one-workgroup-per-query reduction: every width
    (16 / 32 / 64 columns, ragged counts) against the tile route of the same library and, for a sample, against the oracle
    (explicit-inverse route and the tighter triangular-solve route); then the single query with second-order outputs
    (the same kernels in linearize mode: columns [k*, dk*/dx], dot products with column 0)."""
    N = 5000
    syn = orc.make_synthetic(5, N, 2, 1, 64)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    om = cached_oracle_model(5, N, 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    at = max(mu_atol(om), 1e-12)
    for T in (5, 16, 17, 33, 64):
        mu, var, jac = gp.predict(x[:T], None, True)
        gp.set_small_path(0)
        mu0, var0, jac0 = gp.predict(x[:T], None, True)
        gp.set_small_path(1)
        np.testing.assert_allclose(mu, mu0, rtol=1e-11, atol=at)
        np.testing.assert_allclose(jac, jac0, rtol=1e-11, atol=10 * at)
        np.testing.assert_allclose(var, var0, rtol=0, atol=1e-12)          # two summation orders of the same factor
        assert var.min() > 0
    rmu, rvar, rjac = orc.gp_predict(x[:33], om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    mu, var, jac = gp.predict(x[:33], None, True)
    np.testing.assert_allclose(mu, rmu, rtol=1e-9, atol=at)
    np.testing.assert_allclose(jac, rjac, rtol=1e-9, atol=10 * at)
    np.testing.assert_allclose(var, rvar, rtol=0, atol=1e-9)
    _, cvar = orc.gp_predict_chol(x[:33], om["Z"], om["beta"], om["chol"], om["lengthscale"], om["signal_var"])
    np.testing.assert_allclose(var, cvar, rtol=0, atol=2e-11)
    # second order at one query
    x1 = x[7]
    out = gp.linearize_predict(x1[None, :2], x1[None, 2:], True)
    rjv, rhm = orc.gp_linearize_extras(x1, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"])
    np.testing.assert_allclose(out[0][:, 0], rmu[7], rtol=1e-9, atol=at)
    np.testing.assert_allclose(out[1][:, 0], cvar[7], rtol=0, atol=2e-11)
    np.testing.assert_allclose(out[2], rjac[7], rtol=1e-9, atol=10 * at)
    np.testing.assert_allclose(out[3], rjv, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out[4], rhm, rtol=1e-8, atol=100 * at)


def test_config3_at_the_rollout_count_the_bench_times():
    """BASELINE configs[2] at the size `bench.py --workload c3` times: cart-pole dims (n_s = 4, n_u = 1), N = 5000, H = 15,
    65536 rollouts (983040 step evaluations).  Properties on ALL rows (finite, symmetric, positive definite shape
    matrices, positive GP variances implied by them), the one-call chain against H one-step calls on a strided sample of
    512 rollouts, three rollouts against the oracle chain (factorised on the CPU), and bit-equality of a slice evaluated
    on its own (same kernels, same tile positions)."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, workload
    N, T, H, n_s, n_u = 5000, 65536, 15, 4, 1
    syn = orc.make_synthetic(3, N, n_s, n_u, 8, sf2=0.01)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    roll = workload.random_rollout_controls(34, T, H, n_s, n_u)
    dev = gp.device
    tp0, tkff, tkfb = (torch.from_numpy(roll[k]).to(dev) for k in ("p0", "k_ff", "k_fb"))
    l = np.full(n_s, 0.05)
    a, b = 0.5 * np.eye(n_s), np.zeros((n_s, n_u))
    p_all, q_all = reach.multistep_reachability_batch(tp0, gp, tkfb, tkff, l, l, None, 2.0, a, b)
    assert p_all.shape == (T, H, n_s) and q_all.shape == (T, H, n_s, n_s)
    assert bool(torch.isfinite(p_all).all()) and bool(torch.isfinite(q_all).all())
    assert float((q_all - q_all.transpose(-1, -2)).abs().max()) <= 1e-14 * float(q_all.abs().max())
    # positive definite everywhere: batched Cholesky of all 983040 shape matrices succeeds
    _, info = torch.linalg.cholesky_ex(q_all.reshape(-1, n_s, n_s))
    assert int(info.abs().max()) == 0
    # the chain == repeated one-step calls, on every 128th rollout
    idx = torch.arange(0, T, 128, device=dev)
    sp, skff, skfb = tp0[idx], tkff[idx], tkfb[idx]
    p, q = reach.onestep_reachability_batch(sp, gp, skff[:, 0], l, l, None, None, 2.0, a, b)
    # (the 512-query calls take the split-K posterior kernels, the 65536-query chain the plain tiles: another order of
    #  summation, compounded over the steps)
    np.testing.assert_allclose(p.cpu().numpy(), p_all[idx, 0].cpu().numpy(), rtol=1e-9, atol=1e-12)
    for i in range(1, H):
        p, q = reach.onestep_reachability_batch(p, gp, skff[:, i], l, l, q, skfb[:, i - 1], 2.0, a, b)
        np.testing.assert_allclose(p.cpu().numpy(), p_all[idx, i].cpu().numpy(), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(q.cpu().numpy(), q_all[idx, i].cpu().numpy(), rtol=1e-7, atol=1e-14)
    # a tile-aligned slice on its own: bit-equal
    sl = slice(128 * 100, 128 * 108)
    ps, qs = reach.multistep_reachability_batch(tp0[sl], gp, tkfb[sl], tkff[sl], l, l, None, 2.0, a, b)
    np.testing.assert_allclose(qs.cpu().numpy(), q_all[sl].cpu().numpy(), rtol=1e-9, atol=1e-15)
    # oracle chain on three rollouts
    om = cached_oracle_model(3, N, n_s, n_u, sf2=0.01)
    pick = [0, 31337, T - 1]
    rp, rq = orc.multistep_reachability_batch(om, roll["p0"][pick], roll["k_fb"][pick], roll["k_ff"][pick], l, l, None,
                                              2.0, a, b)
    ti = torch.tensor(pick, device=dev)
    np.testing.assert_allclose(p_all[ti].cpu().numpy(), rp, rtol=1e-8, atol=max(mu_atol(om), 1e-11))
    np.testing.assert_allclose(q_all[ti].cpu().numpy(), rq, rtol=1e-6, atol=1e-13)


def test_config3_with_the_parameters_the_survey_wrote():
    """BASELINE configs[2] with SURVEY 8(d)'s parameters for row C3 AS WRITTEN: sigma_f^2 = 1, prior a = I (the shipped `c3`
    workload uses sigma_f^2 = 0.01, a = 0.5 I).  With them the chain is not computable in fp64, by the
    reference either: far from the data the remainder box grows as l_mu r^2 with r^2 = lambda_max(Q (I + K^T K)), so
    Q_{k+1} ~ n_s l_mu^2 |Q_k|^2 = 0.01 |Q_k|^2 -- doubly exponential once |Q| > 100 (trace 0.13, 1.3, 7, 50, 480, 1e4, 1e6,
    6e9, 1e17, 3e31, 2e60, 1e118, 3e233, inf).  Pinned here at the full size (65536 rollouts x 15 steps, N = 5000):
      * every rollout is finite through step 10 and holds its first inf / NaN at step 11, 12, 13 or 14;
      * three rollouts agree with the oracle chain on every step the oracle can compute, and the oracle -- like the
        reference, whose scipy.linalg.eig refuses non-finite input (utils.py:108-144) -- raises ValueError at the step after
        the first inf, the step this build reports;
      * with check_bounds=True the batched entry point raises (the reference's assert u_b > 0, utils_ellipsoid.py:227)."""
    import torch
    from safe_exploration_amd import gp_reachability as reach, workload
    N, T, H, n_s, n_u = 5000, 65536, 15, 4, 1
    syn = orc.make_synthetic(3, N, n_s, n_u, 8, sf2=1.0)
    gp = hip_model(syn["Z"], syn["Y"], syn["lengthscale"], syn["signal_var"], syn["noise_var"], n_s, n_u)
    roll = workload.random_rollout_controls(34, T, H, n_s, n_u)
    dev = gp.device
    tp0, tkff, tkfb = (torch.from_numpy(roll[k]).to(dev) for k in ("p0", "k_ff", "k_fb"))
    l = np.full(n_s, 0.05)
    a, b = np.eye(n_s), np.zeros((n_s, n_u))
    p_all, q_all = reach.multistep_reachability_batch(tp0, gp, tkfb, tkff, l, l, None, 2.0, a, b)
    fin = torch.isfinite(q_all).reshape(T, H, -1).all(-1) & torch.isfinite(p_all).all(-1)
    assert bool(fin[:, :11].all()), "a rollout left fp64 before step 11"
    first = torch.where(fin.all(1), torch.full((T,), H, device=dev), (~fin).int().argmax(1))
    counts = {int(k): int((first == k).sum()) for k in torch.unique(first)}
    print("first non-finite step -> rollouts:", counts)
    assert int(first.min()) >= 11 and int(first.max()) <= 14, counts
    assert not bool(fin[:, H - 1].any()), "a rollout survived the horizon"
    # once gone, gone: no step after the first non-finite one is finite again
    assert bool((fin.int().diff(dim=1) <= 0).all())
    # growth as derived: the trace at step 9 is beyond 1e12 for every rollout (7.6e16 ... 1e65 on this data) and at least
    # squares from there on while it is finite
    tr = q_all.diagonal(dim1=-2, dim2=-1).sum(-1)
    assert float(tr[:, 9].min()) > 1e12
    ok10 = fin[:, 10]
    assert bool((tr[ok10, 10] > 1e-3 * tr[ok10, 9] ** 2).all())
    om = cached_oracle_model(3, N, n_s, n_u, sf2=1.0)
    pick = [0, 31337, T - 1]
    for r in pick:
        p, q = roll["p0"][r:r + 1], None
        raised = None
        with np.errstate(all="ignore"):
            for i in range(H):
                try:
                    p, q, _ = orc.onestep_reachability_batch(om, p, q, roll["k_ff"][r:r + 1, i],
                                                             None if i == 0 else roll["k_fb"][r:r + 1, i - 1], l, l, 2.0, a, b)
                except ValueError:
                    raised = i
                    break
                if np.all(np.isfinite(q)):
                    # (the relative error doubles per step as the magnitudes square: 1e-13 2^13 and the conditioning)
                    np.testing.assert_allclose(p_all[r, i].cpu().numpy(), p[0], rtol=1e-7, atol=1e-9)
                    np.testing.assert_allclose(q_all[r, i].cpu().numpy(), q[0].real, rtol=1e-5, atol=0)
                else:
                    assert int(first[r]) == i, (int(first[r]), i)
        assert raised is not None and raised == int(first[r]) + 1, (raised, int(first[r]))
    with pytest.raises(AssertionError):
        reach.multistep_reachability_batch(tp0[:256], gp, tkfb[:256], tkff[:256], l, l, None, 2.0, a, b, check_bounds=True)


@pytest.mark.parametrize("adds", [[1], [16], [50], [128], [1, 16, 50, 128, 3]])
def test_row_append_at_the_headline_model_size(adds):
    """update_model(replace_old=False) at N = 5000 (where DESIGN quotes 0.40 / 0.73 / 1.2 / 2.25 ms): + 1, + 16, + 50, + 128
    points and a sequence that crosses the 128-padding boundary, against a refit on all the data (posterior AND factor)
    and the oracle's variance on the same data."""
    N0 = 5000
    ntot = N0 + sum(adds)
    syn = orc.make_synthetic(50 + ntot, ntot, 2, 1, 256)
    Z, Y = syn["Z"], syn["Y"]
    gp = hip_model(Z[:N0], Y[:N0], syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    lo = N0
    for m in adds:
        gp.update_model(Z[lo:lo + m], Y[lo:lo + m], opt_hyp=False, replace_old=False)
        lo += m
    assert gp._handle.N == ntot
    full = hip_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"], 2, 1)
    x = np.hstack((syn["p"], syn["k_ff"]))
    mu_a, var_a, jac_a = gp.predict(x, None, True)
    mu_f, var_f, jac_f = full.predict(x, None, True)
    scale = np.abs(full.beta).sum(0).max()
    np.testing.assert_allclose(gp.beta, full.beta, rtol=1e-6, atol=1e-8 * np.abs(full.beta).max())
    np.testing.assert_allclose(mu_a, mu_f, rtol=1e-8, atol=1e-10 * scale)
    np.testing.assert_allclose(jac_a, jac_f, rtol=1e-8, atol=1e-9 * scale)
    np.testing.assert_allclose(var_a, var_f, rtol=0, atol=1e-9)
    np.testing.assert_allclose(gp.predict(x[:1])[1], var_f[:1], rtol=0, atol=1e-9)     # streaming single-query path
    _, wt_a = gp.export_state()
    _, wt_f = full.export_state()
    assert wt_a.shape == wt_f.shape
    den = float(wt_f.abs().max())
    assert float((wt_a - wt_f).abs().max()) <= 1e-8 * den
    assert float(torch_tril_abs_max(wt_a)) == 0.0
    if len(adds) == 1:          # one oracle fit (N^3 on the CPU) per size is enough
        om = oracle_model(Z, Y, syn["lengthscale"], syn["signal_var"], syn["noise_var"])
        _, rvar = orc.gp_predict(x, om["Z"], om["beta"], om["inv_K"], om["lengthscale"], om["signal_var"], False)
        np.testing.assert_allclose(var_a, rvar, rtol=0, atol=1e-9)


def torch_tril_abs_max(w):
    import torch
    return torch.tril(w, -1).abs().max()


def _bench_line(argv, timeout=1500):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, cwd=root, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, "bench.py %s failed:\n%s\n%s" % (argv, r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("argv,N,T,H,n_s", [
    (["--workload", "c2"], 2000, 65536, 1, 2),
    (["--workload", "c2p"], 5000, 65536, 1, 2),
    (["--workload", "c3"], 5000, 65536, 15, 4),
    (["--workload", "c5", "--queries", "196608"], 5000, 196608, 1, 2),
])
def test_bench_lines_are_self_consistent(argv, N, T, H, n_s):
    """`bench.py --workload ...` with one timed step: the JSON line carries the contract's fields and its roofline is
    arithmetic on what was measured -- achieved x avg_launch_ms == flops_per_launch, launches x flops_per_launch == the
    algorithmic flops of the step (n_out N^2 T H), value == evals / time, dominant kernel <= step time."""
    line = _bench_line(argv + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_kstar"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["config"]["N"] == N and line["config"]["queries_per_gpu_per_step"] == T and line["config"]["horizon"] == H
    rf = line["roofline"]
    assert rf["kernel"] == "sr_var_kernel" and rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s"
    flops_step = float(n_s) * N * N * T * H
    assert abs(rf["flops_per_launch"] * rf["launches"] - flops_step) <= 1e-9 * flops_step
    assert abs(rf["achieved"] * 1e12 * rf["avg_launch_ms"] * 1e-3 - rf["flops_per_launch"]) <= 1e-6 * rf["flops_per_launch"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.5 < rf["frac"] < 1.0
    assert abs(rf["frac_of_measured_ceiling"] - rf["achieved"] / rf["measured_ceiling"]) < 1e-12
    assert rf["avg_launch_ms"] * rf["launches"] <= line["ms_per_step"] * 1.001
    assert abs(line["value"] - T * H / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert abs(rf["hbm_gbps_achieved"] - rf["hbm_bytes_per_step_algorithmic"] / (line["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * rf["hbm_gbps_achieved"]
    rk = line["roofline_kstar"]
    assert rk["bound"] == "hbm-write" and 0.2 < rk["frac"] < 1.0
    assert abs(rk["achieved"] * 1e9 * rk["avg_launch_ms"] * 1e-3 - rk["bytes_per_launch"]) <= 1e-6 * rk["bytes_per_launch"]


def test_bench_model_update_line_is_self_consistent():
    """`bench.py --workload c4 --n-train 5000`: value == (2/3) N^3 n_out / step time; per-kernel sums present."""
    line = _bench_line(["--workload", "c4", "--n-train", "5000", "--steps", "3", "--warmup", "2"])
    N, n_out = 5000, 2
    flops = n_out * (2.0 / 3.0) * float(N) ** 3
    assert abs(line["roofline"]["flops_per_step"] - flops) <= 1e-9 * flops
    assert abs(line["value"] - flops / (line["ms_per_step"] * 1e-3) / 1e12) <= 1e-6 * line["value"]
    assert line["unit"] == "TFLOP/s" and 0.2 < line["roofline"]["frac"] < 1.0
    assert line["config"]["max|mu(z)+s2n*alpha-y|"] < 1e-7
    assert set(line["kernel_ms_per_step"]) == {"sr_gram_kernel", "sr_potrf_diag_kernel", "sr_gemm_tn_kernel[cholesky]",
                                               "sr_gemm_tn_kernel[inverse]"}


def test_no_access_past_a_buffer_under_guard_allocation():
    """SR_GUARD=1 makes every device buffer of the library end at the end of its own 2 MiB-granular allocation: an access
    past a buffer leaves the mapping and kills the process instead of landing silently in a neighbour.  (That is how the
    four-rows-past-U^-1 read of the 16-wavefront streaming kernels showed itself -- odd padded sizes, last output.)  With
    SR_POISON=1 the new buffers also hold NaN patterns instead of zeros (the block cache hands out used memory as it is).
    The shape-heavy parity tests run in a child process under that mode and must pass."""
    import subprocess
    env = dict(os.environ, SR_GUARD="1", SR_POISON="1")     # ... and new buffers hold NaN patterns, not zeros
    sel = ("fused_small_model_linearize or ragged or small_batch_streaming or splitk or all_state_action or "
           "row_append or streamed_linearize or fused_small_model_pass or persistent_chain_matches or "
           "resident_server_answers or one_point_append")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
