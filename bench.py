#!/usr/bin/env python3
"""bench.py -- one-step reachability throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2p|c2|c3|c5|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks
(torch.multiprocessing.spawn, one per GPU, rendezvous on 127.0.0.1); under torch.distributed.run the
ranks it finds are used as they are.  Either way the backend is "nccl" (= RCCL on ROCm).  Two environment
overrides exist so that the N > 1 branch can be exercised on a ONE-GPU box (RCCL itself refuses two ranks on one
device): SR_DIST_BACKEND=gloo picks the process-group backend, SR_SHARE_DEVICE=1 puts every rank on cuda:0.  The
JSON line names the backend and device placement that really ran (`config.parallelism`).

One *step* = one pass of the hot path over one batch of T synthetic query states per GPU:
GP posterior (mu, var, d mu/dx at z=[p;k_ff]) + ellipsoid branch of onestep_reachability, through
the C-ABI (sr_onestep_reach).  Inputs are resident in HBM before the timed region.  Default
workload "c2p" is the configuration the metric is quoted on: pendulum (n_s=2, n_u=1), N=5000
training points, T=65536 query states per GPU per step, fp64 (SURVEY.md 8(d) row C2').

Multi-GPU: weak scaling, one process per GPU, the query batch is sharded; rank 0 factorises and
broadcasts the training set + cached posterior (alpha, U^-1) ONCE over RCCL; no data-path collective.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel sr_var_kernel, fp64 MFMA bound, timed
live with hipEvents on the launch stream) and `cpu_baseline` (oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "one-step reachability evals/sec (batched query states), N=5k train pts"
FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X fp64 matrix (v_mfma_f64_16x16x4_f64) dense peak, vendor spec
FP64_MFMA_MEASURED_TFLOPS = 77.4 # what scripts/mfma_f64_peak.hip reaches on this part (2 waves/SIMD, no operands from memory)
HBM_PEAK_GBS = 8000.0
XGMI_LINK_GBS = 153.0            # one xGMI link of an MI355X, per direction (7 links per GPU, point to point): what a ring /
                                 # chain broadcast between two GPUs is bound by

WORKLOADS = {
    # name: (description, seed, N, n_s, n_u, T per GPU, H, signal variance, prior a = a_scale * I)
    "c2p": ("C2' pendulum n_s=2 n_u=1 D=3, N=5000 train pts, T=65536 query states/GPU/step, "
            "one-step ellipsoid branch, fp64", 5, 5000, 2, 1, 65536, 1, 1.0, 1.0),
    "c2": ("C2 pendulum n_s=2 n_u=1 D=3, N=2000 train pts, T=65536 query states/GPU/step, "
           "one-step ellipsoid branch, fp64", 2, 2000, 2, 1, 65536, 1, 1.0, 1.0),
    "c5": ("C5 pendulum n_s=2 n_u=1 D=3, N=5000 train pts, T=1048576 query states/GPU/step (8M over 8 GPUs), "
           "16 chunks of 65536 through the bounded workspace, fp64", 5, 5000, 2, 1, 1048576, 1, 1.0, 1.0),
    # the 15-step chain is only numerically meaningful for a contracting prior model and a GP that
    # models a small residual (sigma_f^2 = 0.01, a = 0.5 I), as in the golden chain fixtures
    "c3": ("C3 cart-pole n_s=4 n_u=1 D=5, N=5000 train pts, T=65536 rollouts/GPU/step, H=15 multi-step "
           "(evals = T*H), sigma_f^2=0.01, prior a=0.5 I, fp64 [SURVEY 8(d) writes sigma_f^2=1, a=I for this row: with those the "
           "shape matrices grow as Q_{k+1} ~ 0.01 Q_k^2 and leave fp64 at step 12-13 of 15, where the reference's scipy.linalg.eig "
           "raises ValueError -- workload c3s runs exactly that, tests/test_gpu_fullsize.py::test_config3_with_the_parameters_"
           "the_survey_wrote pins it]", 3, 5000, 4, 1, 65536, 15, 0.01, 0.5),
    # C3 with the parameters SURVEY 8(d) wrote (sigma_f^2 = 1, a = I): the same flops, a chain that overflows fp64 (see c3)
    "c3s": ("C3 AS SURVEYED: cart-pole n_s=4 n_u=1 D=5, N=5000 train pts, T=65536 rollouts/GPU/step, H=15 multi-step "
            "(evals = T*H), sigma_f^2=1, prior a=I, fp64; the shape matrices overflow fp64 inside the horizon "
            "(config.first_nonfinite_step), every step is evaluated and timed all the same", 3, 5000, 4, 1, 65536, 15, 1.0, 1.0),
}
L_CONST = {2: np.array([0.05, 0.02]), 4: np.array([0.05] * 4)}   # environments.py:317-318, 702-704
C_SAFETY = 2.0                                                     # defaultconfig_exploration.py:35


def cpu_baseline(prob, l_mu, l_sigma, budget_s=12.0):
    """Oracle ('port' of the reference route: explicit-inverse variance, vectorised over the sample)
    timed on this box's host cores.  The model fit is excluded (as for the GPU), the sample is bounded."""
    from oracle import oracle_np as orc
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    n_s = len(prob["signal_var"])
    noise = prob["noise_var"] + 1e-5
    t0 = time.time()
    beta, inv_K, _ = orc.gp_fit(prob["Z"], prob["Y"], prob["lengthscale"], prob["signal_var"], noise)
    fit_s = time.time() - t0
    model = dict(Z=prob["Z"], beta=beta, inv_K=inv_K, lengthscale=prob["lengthscale"],
                 signal_var=prob["signal_var"])
    a, b = np.eye(n_s), np.zeros((n_s, prob["k_ff"].shape[1]))
    Ts, done, spent = 256, 0, 0.0
    while True:
        sl = slice(0, Ts)
        t0 = time.time()
        orc.onestep_reachability_vectorised(model, prob["p"][sl], prob["Q"][sl], prob["k_ff"][sl],
                                            prob["k_fb"][sl], l_mu, l_sigma, C_SAFETY, a, b)
        dt = time.time() - t0
        done, spent = Ts, dt
        if dt > budget_s / 3 or Ts * 2 > prob["p"].shape[0]:
            break
        Ts *= 2
    # how the reference is actually driven (gp_reachability.py:199-210): one Python call per query
    nq = 32
    t0 = time.time()
    for t in range(nq):
        z = np.hstack((prob["p"][t], prob["k_ff"][t]))
        mu, var, jac = orc._predict_one(model, z)
        orc.onestep_reachability_from_gp(prob["p"][t], prob["Q"][t], prob["k_ff"][t], prob["k_fb"][t], mu, var,
                                         jac, l_mu, l_sigma, C_SAFETY, a, b)
    loop_rate = nq / (time.time() - t0)
    return {"value": done / spent, "unit": "evals/s", "cores": int(thr), "kind": "port",
            "sample": "first %d of the same T query states, N=%d; vectorised NumPy/SciPy oracle "
                      "(explicit inv_K route of the reference), %.1f s timed; model fit (%.1f s) excluded"
                      % (done, prob["Z"].shape[0], spent, fit_s),
            "per_query_loop_value": loop_rate,
            "per_query_loop_note": "reference-style driving pattern: one onestep call per query (%d queries, "
                                   "BLAS threads as above)" % nq}


def cpu_baseline_chain(prob, roll, l_mu, l_sigma, a, b, H, budget_s=12.0):
    """C3: the oracle's H-step chain (one reference-style onestep call per rollout and step, gp_reachability.py:195-210)
    on a bounded number of rollouts; value in step-evals/s like the GPU line.  Model fit excluded."""
    from oracle import oracle_np as orc
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    t0 = time.time()
    beta, inv_K, _ = orc.gp_fit(prob["Z"], prob["Y"], prob["lengthscale"], prob["signal_var"], prob["noise_var"] + 1e-5)
    fit_s = time.time() - t0
    model = dict(Z=prob["Z"], beta=beta, inv_K=inv_K, lengthscale=prob["lengthscale"], signal_var=prob["signal_var"])
    n, done, spent = 2, 0, 0.0
    while True:
        t0 = time.time()
        orc.multistep_reachability_batch(model, roll["p0"][:n], roll["k_fb"][:n], roll["k_ff"][:n], l_mu, l_sigma, None,
                                         C_SAFETY, a, b)
        dt = time.time() - t0
        done, spent = n, dt
        if dt > budget_s / 3 or n * 2 > roll["p0"].shape[0]:
            break
        n *= 2
    return {"value": done * H / spent, "unit": "evals/s", "cores": int(thr), "kind": "port",
            "sample": "first %d of the same rollouts x H=%d, N=%d; NumPy/SciPy oracle, one onestep call per rollout "
                      "and step as the reference drives it; %.1f s timed; model fit (%.1f s) excluded"
                      % (done, H, prob["Z"].shape[0], spent, fit_s)}


def cpu_baseline_update(n_s, n_u, N, budget_s=20.0):
    """C4: the oracle's model update (the reference route: Cholesky, explicit inverse and alpha per output with SciPy /
    LAPACK on the host cores) on a bounded number of training points; value in TFLOP/s on the same (2/3) N^3 n_out."""
    from oracle import oracle_np as orc
    from safe_exploration_amd import workload
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        thr = os.cpu_count() or 1
    Ns, done, spent = min(N, 1000), 0, 0.0
    while True:
        prob = workload.make_problem(4, Ns, n_s, n_u, 16)
        t0 = time.time()
        orc.gp_fit(prob["Z"], prob["Y"], prob["lengthscale"], prob["signal_var"], prob["noise_var"] + 1e-5)
        dt = time.time() - t0
        done, spent = Ns, dt
        if dt > budget_s / 4 or Ns * 2 > N:
            break
        Ns *= 2
    flops = n_s * (2.0 / 3.0) * float(done) ** 3
    return {"value": flops / spent / 1e12, "unit": "TFLOP/s", "cores": int(thr), "kind": "port",
            "sample": "model update of N=%d training points (n_out=%d) by the NumPy/SciPy oracle (Cholesky + explicit inverse per "
                      "output), %.2f s timed" % (done, n_s, spent)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2p", choices=sorted(WORKLOADS) + ["c4"])
    ap.add_argument("--queries", type=int, default=0, help="override T per GPU")
    ap.add_argument("--n-train", type=int, default=0, help="override the number of training points")
    ap.add_argument("--var-group", type=int, default=0)
    ap.add_argument("--var-variant", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-nccl", action="store_true", help="with ONE rank: create the nccl (= RCCL) process group all the "
                    "same and send the model through the replication path to itself (object broadcast, tensor broadcasts, "
                    "the packed factor in 64 MB pieces) plus one timed 64 MB broadcast -- so that the RCCL library is loaded and "
                    "has moved bytes on this box before a multi-GPU lease does it for the first time")
    ap.add_argument("--c4-replicas", action="store_true", help="--workload c4 with N > 1 ranks: every rank updates its own "
                    "replica (weak scaling, no exchange) instead of sharding the OUTPUTS over the ranks "
                    "(parallel.fit_outputs_sharded: strong scaling, every factor broadcast once from its owner)")
    ap.add_argument("--dump-shards", default="", help="directory: every rank stores its query seed and the head of its "
                    "outputs of the last step there (shard_rank<r>.npz) -- for tests of the sharded path")
    return ap.parse_args(argv)


def dist_backend():
    return os.environ.get("SR_DIST_BACKEND", "nccl")


def share_device():
    return os.environ.get("SR_SHARE_DEVICE", "0") not in ("", "0")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned_rank(local_rank, argv, world, port):
    """One rank of a self-launched run: the environment torch.distributed.run would have provided."""
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (RCCL)
    run(parse_args(argv))


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU
        import torch
        import torch.multiprocessing as mp
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
        have = torch.cuda.device_count()
        if have < args.gpus and not share_device():
            raise SystemExit("--gpus %d but only %d device(s) visible" % (args.gpus, have))
        mp.spawn(_spawned_rank, args=(list(sys.argv[1:] if argv is None else argv), args.gpus, _free_port()),
                 nprocs=args.gpus, join=True)
        return
    run(args)


def gather_rank_stats(vals, dev, world):
    """(world, len(vals)) array of every rank's numbers, on every rank (one all_gather of a small fp64 tensor)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return np.array([[float(v) for v in vals]])
    # (gloo gathers host tensors only; nccl = RCCL device tensors only)
    mine = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return torch.stack(parts).cpu().numpy()


def rank_section(per_rank_ms):
    """The per-rank part of an N > 1 line: a slow rank, a straggling clock or an uneven shard must be readable from the
    line itself (the scaling run is launched by the driver, nobody watches it)."""
    per = [float(x) for x in per_rank_ms]
    return {"ms_per_step": per, "ms_per_step_min": min(per), "ms_per_step_max": max(per),
            "slowest_rank": int(np.argmax(per)), "spread": (max(per) - min(per)) / max(min(per), 1e-12)}


def run_model_update_sharded(args, dev, world, rank):
    """--workload c4 with N > 1 ranks (default): ONE model update with the outputs dealt to the ranks
    (parallel.fit_outputs_sharded, SURVEY 8(e) / DESIGN 5): rank r factorises outputs r, r + world, ... with no
    communication, every factor then travels once from its owner as its packed upper triangle.  Strong scaling: the
    work is fixed (n_out = 2: two ranks have something to factorise, the others only receive).  value = algorithmic
    TFLOP/s of the whole job over the slowest rank's time, the broadcast included."""
    import torch
    import torch.distributed as dist
    from safe_exploration_amd import workload, parallel
    N = args.n_train or 50000
    n_s, n_u = 2, 1
    prob = workload.make_problem(4, N, n_s, n_u, 16)
    hyp = workload.hyp_list(prob)
    gp = None
    for _ in range(max(args.warmup, 1)):
        gp = None
        gp = parallel.fit_outputs_sharded(n_s, n_s, n_u, prob["Z"], prob["Y"], hyp=hyp, device=dev)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    moved = 0
    for _ in range(args.steps):
        gp = None                                          # (the previous full model goes back to the block cache first)
        gp = parallel.fit_outputs_sharded(n_s, n_s, n_u, prob["Z"], prob["Y"], hyp=hyp, device=dev)
        moved += parallel.LAST_REPLICATION.get("factor_bytes", 0) + parallel.LAST_REPLICATION.get("other_bytes", 0)
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    dist.barrier()
    elapsed = time.perf_counter() - t0
    stats = gather_rank_stats([local, elapsed], dev, world)
    elapsed = float(stats[:, 1].max())
    idx = np.random.default_rng(0).choice(N, min(N, 1024), replace=False)
    s2n = prob["noise_var"] + 1e-5 + 1e-8
    mu, _ = gp.predict(prob["Z"][idx])
    res_mu = float(np.abs(mu + s2n[None, :] * gp.beta[idx] - prob["Y"][idx]).max())
    assert res_mu < 1e-7, "posterior identity violated on rank %d: %g" % (rank, res_mu)
    if rank == 0:
        flops = n_s * (2.0 / 3.0) * float(N) ** 3
        per = elapsed / args.steps
        owners = min(world, n_s)
        line = {
            "metric": "GP model update TFLOP/s (blocked fp64 Cholesky + explicit triangular inverse), N=%d train pts" % N,
            "value": flops / per / 1e12, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * per, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 pendulum dims n_out=2, N=%d training points, ONE model update with the outputs sharded "
                                   "over the ranks (parallel.fit_outputs_sharded), every factor broadcast once from its owner as "
                                   "its packed upper triangle, fp64" % N,
                       "N": N, "n_out": n_s,
                       "parallelism": "output-shard x%d of %d ranks (%s world=%d%s)" % (
                           owners, world, dist.get_backend(), world, ", all ranks on cuda:0" if share_device() else ", one GPU per rank"),
                       "broadcast_bytes_per_step": moved // max(args.steps, 1),
                       "max|mu(z)+s2n*alpha-y|": res_mu},
            "ranks": rank_section(1e3 * stats[:, 0] / args.steps),
            "roofline": {"kernel": "sr_gemm_tn_kernel", "bound": "mfma", "achieved": flops / per / 1e12 / owners,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / per / 1e12 / owners / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "note": "per factorising GPU (%d of the %d ranks own an output), the broadcast inside the time" % (owners, world)},
        }
        print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def run_model_update(args, dev, world, rank):
    """--workload c4 (BASELINE configs[3]): one *step* = one model update of a GP with N training points and
    n_out = 2 outputs -- Gram matrix, blocked fp64 Cholesky with MFMA trailing updates, explicit U^-1, alpha --
    through sr_gp_set_data + sr_gp_factorize.  N = 50000 unless --n-train says otherwise (the 40 GB of factors stay
    resident in HBM).  value = algorithmic TFLOP/s, (2/3) N^3 n_out per update (potrf N^3/3 + trtri N^3/3).
    Every rank updates its own replica of the model (the path has no exchange step); value sums over ranks."""
    import torch
    import torch.distributed as dist
    from safe_exploration_amd import SimpleGPModel, workload, _lib
    N = args.n_train or 50000
    n_s, n_u = 2, 1
    prob = workload.make_problem(4, N, n_s, n_u, 16)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device=dev)
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)          # first update allocates the factors and the scratch
    for _ in range(max(args.warmup, 1)):                   # (and touches 120 GB for the first time at N = 50000)
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)      # same shape: refactorises in place
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel times from ONE more, untimed update: the hipEvent pair around each of the ~300 launches of an update
    # costs 1.5 ms at N = 5000 (8.2 against 6.7 ms) -- negligible at N = 50000, but not part of the path
    gp.prof_reset()
    gp.prof_enable(True)
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize(dev)
    gp.prof_enable(False)
    prof_steps = 1
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # posterior identities at the training inputs (no CPU oracle reaches N = 50000): K_y alpha = y
    idx = np.random.default_rng(0).choice(N, min(N, 1024), replace=False)
    s2n = prob["noise_var"] + 1e-5 + 1e-8
    mu, var = gp.predict(prob["Z"][idx])
    res_mu = float(np.abs(mu + s2n[None, :] * gp.beta[idx] - prob["Y"][idx]).max())
    assert res_mu < 1e-7, "posterior identity violated: %g" % res_mu
    if rank == 0:
        flops = n_s * (2.0 / 3.0) * float(N) ** 3
        per = elapsed / args.steps
        ms = {k: gp.prof_get(i) for k, i in (("sr_gram_kernel", _lib.K_GRAM), ("sr_potrf_diag_kernel", _lib.K_POTRF),
                                              ("sr_gemm_tn_kernel[cholesky]", _lib.K_GEMM),
                                              ("sr_gemm_tn_kernel[inverse]", _lib.K_TRINV))}
        gemm_ms = ms["sr_gemm_tn_kernel[cholesky]"][0] + ms["sr_gemm_tn_kernel[inverse]"][0]
        line = {
            "metric": "GP model update TFLOP/s (blocked fp64 Cholesky + explicit triangular inverse), N=%d train pts" % N,
            "value": world * flops / per / 1e12, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * per, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4 pendulum dims n_out=2, N=%d training points, model update "
                                   "(sr_gp_set_data + sr_gp_factorize), factors HBM-resident (%.1f GB), fp64"
                                   % (N, n_s * gp._handle.Np ** 2 * 8 / 1e9),
                       "N": N, "n_out": n_s, "parallelism": "replica x%d (no exchange step)" % world,
                       "max|mu(z)+s2n*alpha-y|": res_mu},
            "roofline": {"kernel": "sr_gemm_tn_kernel", "bound": "mfma", "achieved": flops / per / 1e12,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops / per / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "flops_per_step": flops,
                         "note": "achieved = (2/3) N^3 n_out / wall time of the whole update (all kernels, both "
                                 "outputs); kernel_ms_per_step sums hipEvent pairs per launch (outputs overlap on "
                                 "their own streams, so the sum can exceed the wall time)"},
            "kernel_ms_per_step": {k: v[0] / prof_steps for k, v in ms.items()},
            "kernel_launches_per_step": {k: v[1] / prof_steps for k, v in ms.items()},
            "gemm_TFLOPs_over_gemm_ms": flops / max(gemm_ms / prof_steps, 1e-9) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_update(n_s, n_u, N)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run(args):
    if args.var_variant not in (-1, 4) or os.environ.get("SR_FACT_FREE_RATIO"):
        # A/B forms and measurement switches exist in the lab build only (make -C safe_exploration_amd/csrc lab)
        os.environ.setdefault("SAFEREACH_LIB", os.path.join(ROOT, "scripts", "_bin", "libsafereach_lab.so"))
    import torch
    import torch.distributed as dist
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, parallel
    from safe_exploration_amd import _lib, _buffers as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    dev_index = 0 if share_device() else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = "single process"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dist_backend() == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(dist_backend(), rank=rank, world_size=world)
        # what the process group itself reports, and where the ranks sit
        backend = "%s world=%d%s" % (dist.get_backend(), dist.get_world_size(),
                                     ", all ranks on cuda:0" if share_device() else ", one GPU per rank")

    if args.workload == "c4":
        if world > 1 and not args.c4_replicas:
            return run_model_update_sharded(args, dev, world, rank)
        return run_model_update(args, dev, world, rank)
    desc, seed, N, n_s, n_u, T, H, sf2, a_scale = WORKLOADS[args.workload]
    if args.queries:
        T = args.queries
    if args.n_train:
        N = args.n_train
        desc += " [N overridden: %d]" % N
    prob = workload.make_problem(seed, N, n_s, n_u, T, sf2=sf2)     # model part identical on every rank
    a_lin, b_lin = a_scale * np.eye(n_s), np.zeros((n_s, n_u))
    l_mu = l_sigma = L_CONST[n_s]

    # ---- model: rank 0 factorises, the others receive alpha / U^-1 over RCCL --------------------
    t0 = time.time()
    gp = None
    if rank == 0:
        gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device=dev)
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        torch.cuda.synchronize(dev)
    fit_s = time.time() - t0
    bcast_s, bcast = 0.0, {}
    if world > 1:
        dist.barrier()
        t0 = time.time()
        gp = parallel.replicate_model(gp, prob, src=0, device=dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        bcast_s = time.time() - t0
        bcast = dict(parallel.LAST_REPLICATION)
    dry = None
    if args.dry_nccl and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t0 = time.time()
        same = parallel.replicate_model(gp, prob, src=0, device=dev)        # one rank: the sender's half of every step
        assert same is gp
        rep_s = time.time() - t0
        piece = torch.empty(8 << 20, dtype=torch.float64, device=dev).normal_()
        ref = piece.clone()
        dist.broadcast(piece, src=0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            dist.broadcast(piece, src=0)
        torch.cuda.synchronize(dev)
        bc_s = (time.perf_counter() - t0) / 5
        assert bool((piece == ref).all())
        # the collective the N > 1 line takes its per-rank numbers from (gather_rank_stats): device tensors through RCCL
        probe = torch.arange(10, dtype=torch.float64, device=dev)
        got = [torch.empty_like(probe)]
        dist.all_gather(got, probe)
        assert bool((got[0] == probe).all())
        dry = {"backend": dist.get_backend(), "world": dist.get_world_size(), "replication_s": round(rep_s, 4),
               "replication_bytes": parallel.LAST_REPLICATION.get("factor_bytes", 0) + parallel.LAST_REPLICATION.get("other_bytes", 0),
               "pieces": parallel.LAST_REPLICATION.get("pieces"), "broadcast_64MB_ms": round(1e3 * bc_s, 4),
               "broadcast_64MB_GBps": (64 << 20) / bc_s / 1e9, "xgmi_link_GBps": XGMI_LINK_GBS}
        backend = "single process; dry run of the %s process group with one rank" % dist.get_backend()
        dist.destroy_process_group()
    if args.var_group:
        gp.set_var_group(args.var_group)
    if args.var_variant >= 0:
        gp.set_var_variant(args.var_variant)

    # ---- this rank's shard of the (world * T) query states, resident in HBM ------------------------
    q_seed = seed + 7919 + 104729 * rank
    q = workload.make_queries(q_seed, n_s, n_u, T) if rank else prob
    tp, tq, tkff, tkfb = (B.as_dev(q[k], dev) for k in ("p", "Q", "k_ff", "k_fb"))
    if H > 1:
        roll = workload.random_rollout_controls(seed + 31 * rank, T, H, n_s, n_u)
        tp0, tkffH, tkfbH = (B.as_dev(roll[k], dev) for k in ("p0", "k_ff", "k_fb"))

    def step():
        if H == 1:
            return reach.onestep_reachability_batch(tp, gp, tkff, l_mu, l_sigma, tq, tkfb, C_SAFETY, a_lin, b_lin)
        return reach.multistep_reachability_batch(tp0, gp, tkfbH, tkffH, l_mu, l_sigma, None, C_SAFETY, a_lin, b_lin)

    for _ in range(args.warmup):
        step()
    gp.prof_reset()
    gp.prof_enable(True)                 # event pairs on the launch stream, resolved after the region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    local_s = time.perf_counter() - t0   # this rank alone (before the barrier)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gp.prof_enable(False)
    # every rank's own numbers (its time to its own synchronize, the dominant kernel's hipEvent pairs), so that the line is
    # computed from the SLOWEST rank and says how far the ranks are apart; `elapsed` = MAX over ranks of the bracketed time
    stats = gather_rank_stats([elapsed, local_s] + list(gp.prof_get(_lib.K_VAR)) + list(gp.prof_get(_lib.K_KSTAR))
                              + list(gp.prof_get(_lib.K_ELL)) + list(gp.prof_get(_lib.K_FINAL)), dev, world)
    elapsed = float(stats[:, 0].max())
    slowest = int(np.argmax(stats[:, 1]))
    overflow = None
    if args.workload == "c3s":
        # the surveyed parameters leave fp64 inside the horizon (by construction of the chain, not of this build: the
        # reference raises ValueError at the same step): report where, instead of asserting a finite result
        fin = torch.isfinite(out[1]).reshape(out[1].shape[0], H, -1).all(-1)
        first = torch.where(fin.all(1), torch.full((fin.shape[0],), H, device=dev), (~fin).int().argmax(1))
        overflow = [int(first.min()), int(first.median()), int(first.max())]
        assert bool(torch.isfinite(out[1][:, :8]).all()), "non-finite result before the chain can overflow"
    else:
        assert bool(torch.isfinite(out[1]).all()), "non-finite result in the timed region"
    if args.dump_shards:
        os.makedirs(args.dump_shards, exist_ok=True)
        head = min(T, 4096)
        np.savez(os.path.join(args.dump_shards, "shard_rank%d.npz" % rank), rank=rank, world=world, T=T, H=H,
                 query_seed=(-1 if rank == 0 else q_seed), roll_seed=seed + 31 * rank,
                 p_head=out[0][:head].cpu().numpy(), q_head=out[1][:head].cpu().numpy(),
                 p_sum=out[0].sum(0).cpu().numpy(), q_sum=out[1].sum(0).cpu().numpy())

    if rank == 0:
        evals = float(world) * T * H * args.steps
        # the roofline objects are those of the SLOWEST rank (rank 0 at N = 1)
        var_ms, var_n, ks_ms, ks_n, ell_ms, ell_n, fin_ms, fin_n = (float(v) for v in stats[slowest, 2:10])
        # algorithmic flops of one sr_var_kernel launch: n_out * N^2 * T  (N^2/2 MACs per query and
        # output through the triangular factor) -- true N, not the padded one
        # (a launch covers whatever share of the queries the library gave it -- one chunk of <= 65536: the launches of
        #  the timed region together cover T * H * steps queries)
        flops_launch = float(n_s) * N * N * T * H * args.steps / max(var_n, 1)
        avg_ms = var_ms / max(var_n, 1)
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc pass (counters cannot be read from
        # inside the run); the committed summary is quoted and its source named, never presented as live
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % args.workload)
        if os.path.exists(pmc) and not args.n_train and not args.queries:
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("sr_var_kernel_hbm_bytes_per_launch")
                traffic_src = "committed PMC pass: %s (not measured in this run)" % pj.get("source", "profiles/pmc_%s.json" % args.workload)
            except Exception:
                traffic = None
        # SURVEY 8(d) asks for BOTH rates: the algorithmic HBM bytes of a step over the step time (far below the roof
        # here: the batch is MFMA-bound from ~40 queries on) next to the MFMA fraction.  Per 65536-query pass the
        # factor is read once (n_out N (N+1)/2 doubles) plus Z and alpha; per query the API arrays in and out.
        D = n_s + n_u
        passes = H * -(-T // 65536)
        bytes_step = 8.0 * (passes * (n_s * N * (N + 1) / 2 + N * D + n_s * N)
                            + T * H * (n_s + n_u + n_s * n_s + n_u * n_s + n_s + n_s * n_s))
        step_s = elapsed / args.steps
        # the K* pass: bound by its HBM writes (n_out N T doubles per pass: the cross-covariance the contraction reads)
        ks_avg_ms = ks_ms / max(ks_n, 1)
        ks_bytes = 8.0 * n_s * N * T * H * args.steps / max(ks_n, 1)
        ks_gbs = ks_bytes / (ks_avg_ms * 1e-3) / 1e9 if ks_avg_ms > 0 else 0.0
        line = {
            "metric": METRIC, "value": evals / elapsed, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": desc, "N": N, "queries_per_gpu_per_step": T, "horizon": H,
                       "n_s": n_s, "n_u": n_u,
                       "parallelism": "query-shard x%d (%s), one-time broadcast of Z/alpha and the packed upper "
                                      "triangle of U^-1, no data-path collective" % (world, backend),
                       "model_fit_s": round(fit_s, 3), "broadcast_s": round(bcast_s, 3)},
            "ranks": rank_section(1e3 * stats[:, 1] / args.steps),
            "roofline": {"kernel": "sr_var_kernel", "bound": "mfma", "rank": slowest, "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "frac_of_measured_ceiling": achieved / FP64_MFMA_MEASURED_TFLOPS,
                         "measured_ceiling": FP64_MFMA_MEASURED_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "flops_per_launch": flops_launch, "avg_launch_ms": avg_ms,
                         "launches": var_n,
                         "hbm_gbps_achieved": bytes_step / step_s / 1e9,
                         "hbm_frac": bytes_step / step_s / 1e9 / HBM_PEAK_GBS,
                         "hbm_bytes_per_step_algorithmic": bytes_step,
                         "hbm_note": "SURVEY 8(d) B = factor once per 65536-query pass + Z + alpha + the API arrays; the "
                                     "K* workspace round trip of this design (roofline_kstar.bytes_per_launch written, "
                                     "then read by sr_var_kernel) is extra and still far below the roof"},
            "roofline_kstar": {"kernel": "sr_kstar_kernel", "bound": "hbm-write", "achieved": ks_gbs,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ks_gbs / HBM_PEAK_GBS,
                               "bytes_per_launch": ks_bytes, "avg_launch_ms": ks_avg_ms, "launches": ks_n,
                               "traffic": None},
            "kernel_ms_per_step": {"sr_kstar_kernel": ks_ms / args.steps, "sr_var_kernel": var_ms / args.steps,
                                   "sr_finalize_kernel": fin_ms / args.steps,
                                   "sr_ellipsoid_kernel": ell_ms / args.steps},
            # how busy the GPU is INSIDE the timed region (hipEvent pairs of the four kernels of a step over the wall time of
            # the region): the process as a whole spends most of its wall time in the model fit and the CPU baseline, so a
            # utilisation sampler that looks at it a few times mostly sees an idle device
            "timed_region": {"wall_s": elapsed, "kernel_s": (ks_ms + var_ms + fin_ms + ell_ms) * 1e-3,
                             "gpu_busy_frac": (ks_ms + var_ms + fin_ms + ell_ms) * 1e-3 / elapsed,
                             "rank": slowest},
        }
        if dry is not None:
            line["config"]["dry_nccl"] = dry
        if overflow is not None:
            line["config"]["first_nonfinite_step"] = {"min": overflow[0], "median": overflow[1], "max": overflow[2],
                                                      "note": "0-based step of the chain at which a rollout's shape matrix first "
                                                              "holds an inf / NaN, over this rank's rollouts of the last timed step"}
        if world > 1:
            fb = bcast.get("factor_bytes", 0) + bcast.get("other_bytes", 0)
            line["config"].update({"broadcast_bytes": fb, "broadcast_dense_factor_bytes": bcast.get("dense_factor_bytes"),
                                   "broadcast_pieces": bcast.get("pieces"),
                                   "broadcast_GBps": fb / bcast_s / 1e9 if bcast_s > 0 else None,
                                   "xgmi_link_GBps": XGMI_LINK_GBS,
                                   "broadcast_frac_of_xgmi_link": fb / bcast_s / 1e9 / XGMI_LINK_GBS if bcast_s > 0 else None,
                                   "broadcast_note": "one-time replication (host-side packing and unpacking of the 64 MB pieces "
                                                     "included); a chain / ring broadcast is bound by ONE xGMI link per hop"})
        if world == 1 and not args.no_cpu_baseline:
            if H == 1:
                line["cpu_baseline"] = cpu_baseline(prob, l_mu, l_sigma)
            elif overflow is not None:
                # (the oracle, like the reference, raises where the chain leaves fp64: its first 8 steps are timed)
                short = {"p0": roll["p0"], "k_ff": roll["k_ff"][:, :8], "k_fb": roll["k_fb"][:, :7]}
                line["cpu_baseline"] = cpu_baseline_chain(prob, short, l_mu, l_sigma, a_lin, b_lin, 8)
            else:
                line["cpu_baseline"] = cpu_baseline_chain(prob, roll, l_mu, l_sigma, a_lin, b_lin, H)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
