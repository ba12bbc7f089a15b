"""GPU test of the multi-rank path on ONE device: two processes (gloo rendezvous, CUDA tensors) share
cuda:0; rank 0 factorises, rank 1 adopts the broadcast posterior (sr_gp_import) and evaluates its own
query shard.  The sharded result must equal the single-process result.  (RCCL itself refuses two ranks
on one device; the 8-GPU RCCL run is the driver's scaling bench.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload, parallel
        n_s, n_u, N, T = 2, 1, 300, 1001
        prob = workload.make_problem(21, N, n_s, n_u, T, sf2=0.01)
        gp = None
        if rank == 0:
            gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        gp = parallel.replicate_model(gp, prob, src=0)
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
            ret["shape"] = full_q.shape
        else:
            np.testing.assert_array_equal(gp.beta.shape, (N, n_s))     # imported alpha readable
        dist.barrier()
        # outputs sharded over the ranks for the model update (rank d % world factorises output d)
        n4 = 3
        prob4 = workload.make_problem(22, 260, n4, 1, 64, sf2=0.01)
        hyp4 = workload.hyp_list(prob4)
        sharded = parallel.fit_outputs_sharded(n4, n4, 1, prob4["Z"], prob4["Y"], ["rbf"] * n4, hyp4)
        x4 = np.hstack((prob4["p"], prob4["k_ff"]))
        mu_s, var_s = sharded.predict(x4)
        if rank == 0:
            single = SimpleGPModel(n4, n4, 1, kern_types=["rbf"] * n4, hyp=hyp4)
            single.train(prob4["Z"], prob4["Y"], opt_hyp=False)
            mu_1, var_1 = single.predict(x4)
            ret["d_mu_sharded_fit"] = float(np.abs(mu_s - mu_1).max())
            ret["d_var_sharded_fit"] = float(np.abs(var_s - var_1).max())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu(lib_built):
    assert torch.cuda.is_available()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert ret["shape"] == (1001, 2, 2)
        assert ret["dp"] < 1e-13 and ret["dq"] < 1e-12
        assert ret["d_mu_sharded_fit"] == 0.0 and ret["d_var_sharded_fit"] == 0.0


def _run_bench(argv, env_extra=None, timeout=1500):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, cwd=root,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "bench.py failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def _check_rank_fields(line, world):
    """an N > 1 line must diagnose itself -- every rank's own ms_per_step with min / max / slowest rank,
    the roofline objects computed from the SLOWEST rank, and the one-time broadcast against the xGMI per-link figure."""
    rk = line["ranks"]
    assert len(rk["ms_per_step"]) == world and all(v > 0 for v in rk["ms_per_step"])
    assert rk["ms_per_step_min"] == min(rk["ms_per_step"]) and rk["ms_per_step_max"] == max(rk["ms_per_step"])
    assert 0 <= rk["slowest_rank"] < world and rk["ms_per_step"][rk["slowest_rank"]] == rk["ms_per_step_max"]
    assert rk["spread"] >= 0
    # the bracketed time (barrier on both sides, MAX over ranks) is never shorter than the slowest rank's own
    assert line["ms_per_step"] >= rk["ms_per_step_max"] * (1 - 1e-9)
    assert line["roofline"]["rank"] == rk["slowest_rank"]
    cfg = line["config"]
    assert cfg["xgmi_link_GBps"] == 153.0
    assert abs(cfg["broadcast_frac_of_xgmi_link"] - cfg["broadcast_GBps"] / 153.0) < 1e-12


def test_bench_model_update_with_the_outputs_sharded_over_two_ranks(lib_built):
    """``bench.py --workload c4 --gpus 2``: ONE model update with the two outputs dealt to two ranks
    (parallel.fit_outputs_sharded), every factor broadcast once from its owner as its packed upper triangle -- on the one
    device of this box over gloo, at N = 3000.  Strong scaling: the line says so, counts the flops of the whole job once,
    carries every rank's own time and has checked the posterior identity on every rank."""
    N = 3000
    line = _run_bench(["--gpus", "2", "--workload", "c4", "--n-train", str(N), "--steps", "2", "--warmup", "1"],
                      {"SR_DIST_BACKEND": "gloo", "SR_SHARE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["unit"] == "TFLOP/s"
    assert "output-shard x2 of 2 ranks" in line["config"]["parallelism"] and "gloo" in line["config"]["parallelism"]
    flops = 2 * (2.0 / 3.0) * float(N) ** 3
    assert abs(line["value"] - flops / (line["ms_per_step"] * 1e-3) / 1e12) < 1e-9 * line["value"]
    packed = N * (N + 1) // 2 * 8
    assert line["config"]["broadcast_bytes_per_step"] == 2 * packed + 2 * N * 8
    rk = line["ranks"]
    assert len(rk["ms_per_step"]) == 2 and rk["ms_per_step_max"] <= line["ms_per_step"] * (1 + 1e-9)
    assert line["config"]["max|mu(z)+s2n*alpha-y|"] < 1e-9


def test_bench_two_ranks_c5_on_one_gpu(lib_built, tmp_path):
    """bench.py's N > 1 branch, executed: two self-spawned ranks share cuda:0 over gloo (SR_DIST_BACKEND /
    SR_SHARE_DEVICE; RCCL refuses two ranks on one device), workload c5 (N = 5000, 131072 queries per rank in two
    chunks), packed-triangle replication.  The parsed line must describe what ran, and rank 1's shard must equal a
    single-process evaluation of the same query rows on a model factorised HERE (the replica was never factorised)."""
    T = 131072
    line = _run_bench(["--gpus", "2", "--workload", "c5", "--queries", str(T), "--steps", "2", "--warmup", "1",
                       "--dump-shards", str(tmp_path)],
                      {"SR_DIST_BACKEND": "gloo", "SR_SHARE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    par = line["config"]["parallelism"]
    assert "gloo" in par and "world=2" in par and "all ranks on cuda:0" in par
    assert line["config"]["broadcast_s"] > 0
    N, n_out = 5000, 2
    packed = n_out * N * (N + 1) // 2 * 8
    assert line["config"]["broadcast_bytes"] == packed + (N * 3 + N * n_out + n_out * N) * 8
    assert line["config"]["broadcast_dense_factor_bytes"] == n_out * 5120 * 5120 * 8
    assert line["config"]["broadcast_bytes"] < 0.51 * line["config"]["broadcast_dense_factor_bytes"]
    assert line["config"]["broadcast_GBps"] > 0
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert abs(line["value"] - 2 * T * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    _check_rank_fields(line, 2)
    # rank 1's shard against a single-process evaluation
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload
    d1 = np.load(os.path.join(str(tmp_path), "shard_rank1.npz"))
    assert int(d1["rank"]) == 1 and int(d1["world"]) == 2 and int(d1["T"]) == T
    prob = workload.make_problem(5, N, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob))
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    q = workload.make_queries(int(d1["query_seed"]), 2, 1, T)
    l = np.array([0.05, 0.02])
    head = d1["p_head"].shape[0]
    p1, q1 = reach.onestep_reachability_batch(q["p"], gp, q["k_ff"], l, l, q["Q"], q["k_fb"], 2.0,
                                              np.eye(2), np.zeros((2, 1)))
    np.testing.assert_allclose(d1["p_head"], p1[:head], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d1["q_head"], q1[:head], rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(d1["p_sum"], p1.sum(0), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(d1["q_sum"], q1.sum(0), rtol=1e-10)
    d0 = np.load(os.path.join(str(tmp_path), "shard_rank0.npz"))
    assert int(d0["query_seed"]) == -1 and not np.array_equal(d0["p_head"], d1["p_head"])


def test_bench_eight_ranks_c5_on_one_gpu(lib_built, tmp_path):
    """Rehearsal of BASELINE config 5 as the driver's 8-GPU lease will launch it -- ``bench.py --gpus 8 --workload c5`` --
    on the ONE device of this box: eight self-spawned ranks (gloo rendezvous, SR_SHARE_DEVICE: every replica of the
    N = 5000 model and its 65536-query workspace on cuda:0, about 46 GB in all), one step of 65536 queries per rank.
    Checks what only a world of 8 exercises: the piece schedule of the packed-triangle replication with seven
    receivers, the port / spawn logic, the max-over-ranks timing and the LAST rank's shard against a single-process
    evaluation of the same rows on a model factorised here."""
    T, world = 65536, 8
    free, _total = torch.cuda.mem_get_info(0)
    if free < 60e9:
        pytest.skip("needs ~46 GB of free device memory for eight replicas")
    line = _run_bench(["--gpus", str(world), "--workload", "c5", "--queries", str(T), "--steps", "1", "--warmup", "1",
                       "--dump-shards", str(tmp_path)],
                      {"SR_DIST_BACKEND": "gloo", "SR_SHARE_DEVICE": "1"}, timeout=2400)
    assert line["n_gpus"] == world and line["steps"] == 1 and line["scaling"] == "weak"
    par = line["config"]["parallelism"]
    assert "gloo" in par and "world=8" in par and "all ranks on cuda:0" in par and "query-shard x8" in par
    N, n_out = 5000, 2
    packed = n_out * N * (N + 1) // 2 * 8
    from safe_exploration_amd import parallel
    assert line["config"]["broadcast_pieces"] == n_out * len(parallel.packed_pieces(N))
    assert line["config"]["broadcast_bytes"] == packed + (N * 3 + N * n_out + n_out * N) * 8
    assert line["config"]["broadcast_dense_factor_bytes"] == n_out * 5120 * 5120 * 8
    assert line["config"]["broadcast_s"] > 0 and line["config"]["broadcast_GBps"] > 0
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert abs(line["value"] - world * T / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert "cpu_baseline" not in line                      # rank 0 at N = 1 only
    _check_rank_fields(line, world)
    # every rank wrote its shard; all seeds differ; the last rank's rows equal a single-process evaluation
    from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload
    shards = [np.load(os.path.join(str(tmp_path), "shard_rank%d.npz" % r)) for r in range(world)]
    assert [int(d["rank"]) for d in shards] == list(range(world)) and all(int(d["world"]) == world for d in shards)
    assert len({int(d["query_seed"]) for d in shards}) == world
    d7 = shards[world - 1]
    prob = workload.make_problem(5, N, 2, 1, 16)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob))
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    q = workload.make_queries(int(d7["query_seed"]), 2, 1, T)
    l = np.array([0.05, 0.02])
    head = d7["p_head"].shape[0]
    p1, q1 = reach.onestep_reachability_batch(q["p"], gp, q["k_ff"], l, l, q["Q"], q["k_fb"], 2.0,
                                              np.eye(2), np.zeros((2, 1)))
    np.testing.assert_allclose(d7["p_head"], p1[:head], rtol=0, atol=1e-13)
    np.testing.assert_allclose(d7["q_head"], q1[:head], rtol=1e-12, atol=1e-16)
    np.testing.assert_allclose(d7["p_sum"], p1.sum(0), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(d7["q_sum"], q1.sum(0), rtol=1e-10)


def test_bench_eight_ranks_under_torchrun_default_workload(lib_built, tmp_path):
    """The command line the driver's scaling run uses -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`` with the DEFAULT workload (c2p,
    the headline) -- with the eight ranks sharing cuda:0 over gloo.  bench.py must take RANK / LOCAL_RANK / WORLD_SIZE
    from the environment (no second spawn), and exactly one JSON line must come out."""
    import json
    import subprocess
    import sys
    free, _total = torch.cuda.mem_get_info(0)
    if free < 60e9:
        pytest.skip("needs ~46 GB of free device memory for eight replicas")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SR_DIST_BACKEND="gloo", SR_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2",
           "--warmup", "1", "--dump-shards", str(tmp_path)]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, "torchrun bench failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["metric"].startswith("one-step reachability evals/sec") and line["config"]["N"] == 5000
    assert line["config"]["queries_per_gpu_per_step"] == 65536 and "world=8" in line["config"]["parallelism"]
    assert abs(line["value"] - 8 * 65536 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert sorted(os.listdir(str(tmp_path))) == ["shard_rank%d.npz" % r for r in range(8)]
    _check_rank_fields(line, 8)


def test_bench_dry_run_of_the_rccl_process_group_with_one_rank(lib_built):
    """the first RCCL call must not happen on the multi-GPU lease.  ``bench.py --dry-nccl`` on one GPU
    creates the nccl (= RCCL) process group with a single rank, sends the model through parallel.replicate_model to itself
    (object broadcast, tensor broadcasts, the packed upper triangle in 64 MB pieces through the two staging buffers, async
    works) and times a 64 MB broadcast: librccl is loaded, a communicator is built and bytes move through it."""
    line = _run_bench(["--dry-nccl", "--workload", "c2", "--queries", "8192", "--steps", "1", "--warmup", "1",
                       "--no-cpu-baseline"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    d = line["config"]["dry_nccl"]
    assert d["backend"] == "nccl" and d["world"] == 1
    N, n_out = 2000, 2
    assert d["replication_bytes"] == (n_out * N * (N + 1) // 2 + N * 3 + N * n_out + n_out * N) * 8
    assert d["pieces"] >= 2 and d["broadcast_64MB_ms"] > 0
    assert d["broadcast_64MB_GBps"] > 0 and d["xgmi_link_GBps"] == 153.0
    assert "nccl" in line["config"]["parallelism"]
    assert line["ranks"]["ms_per_step"] and line["ranks"]["slowest_rank"] == 0 and line["roofline"]["rank"] == 0
    assert np.isfinite(line["value"]) and line["value"] > 0
