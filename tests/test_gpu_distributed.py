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
