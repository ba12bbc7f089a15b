"""world_size-2 CPU (gloo) test of the multi-GPU plumbing: query sharding, the one-time model
broadcast and the in-order gather.  The data path itself has no collective."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from safe_exploration_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, T, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        payload = None
        if rank == 0:
            rng = np.random.default_rng(0)
            payload = {"Z": torch.from_numpy(rng.standard_normal((7, 3))),
                       "alpha": torch.from_numpy(rng.standard_normal((2, 7))),
                       "wt": torch.from_numpy(rng.standard_normal((2, 128, 128)))}
        got = parallel.broadcast_tensors(payload, src=0, device=torch.device("cpu"))
        ref = np.random.default_rng(0)
        assert np.array_equal(got["Z"].numpy(), ref.standard_normal((7, 3)))
        assert np.array_equal(got["alpha"].numpy(), ref.standard_normal((2, 7)))
        assert got["wt"].shape == (2, 128, 128) and got["wt"].dtype == torch.float64
        lo, hi = parallel.shard_bounds(T, world, rank)
        local = (np.arange(lo, hi, dtype=np.float64)[:, None] * np.ones((1, 2))) * got["Z"][0, 0].item()
        full = parallel.gather_rows(local, dst=0)
        if rank == 0:
            ret["full"] = full
            ret["scale"] = got["Z"][0, 0].item()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_broadcast_gather_world2():
    T = 11
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), T, ret), nprocs=2, join=True)
        full = ret["full"]
        assert full.shape == (T, 2)
        np.testing.assert_array_equal(full[:, 0], np.arange(T) * ret["scale"])


def _model_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from safe_exploration_amd import SimpleGPModel
        gp = None
        if rank == 0:
            # a model object as rank 0 would hold it after train(m=...): z_fit / y_z are the SELECTED rows, y_train
            # the full data set; the device export is replaced by host tensors (no GPU in this test)
            rng = np.random.default_rng(3)
            hyp = [{"lengthscale": np.array([0.5, 1.0, 1.5]), "variance": 0.7, "noise_variance": 0.02},
                   {"lengthscale": np.array([1.5, 1.0, 0.5]), "variance": 1.3, "noise_variance": 0.03}]
            gp = SimpleGPModel(2, 2, 1, kern_types=["rbf", "mat52"], hyp=hyp)
            gp.x_train, gp.y_train = rng.standard_normal((9, 3)), rng.standard_normal((9, 2))
            gp._z_fit, gp._y_z = gp.x_train[[1, 4, 7, 8, 2]], gp.y_train[[1, 4, 7, 8, 2]]
            gp.z, gp._noise_diag = gp._z_fit, 3e-5
            gp.export_state = lambda: (torch.from_numpy(rng.standard_normal((2, 5))),
                                       torch.from_numpy(rng.standard_normal((2, 128, 128))))
            ret["z"], ret["y"] = gp._z_fit, gp._y_z
        spec, got = parallel.broadcast_model_state(gp, src=0, device=torch.device("cpu"))
        if rank == 1:
            ret["spec"] = spec
            ret["Z"], ret["Y"] = got["Z"].numpy(), got["Y"].numpy()
            ret["shapes"] = (tuple(got["alpha"].shape), tuple(got["wt"].shape))
            # the receiver can rebuild an identical (untrained) model description from the spec alone
            local = SimpleGPModel(spec["n_s_out"], spec["n_s_in"], spec["n_u"], kern_types=spec["kern_types"],
                                  hyp=spec["hyp"])
            ret["noise"] = local._noise.copy()
            ret["ls1"] = local.hyp[1]["lengthscale"]
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_model_state_broadcast_world2():
    """what replicate_model sends: the rows the model is conditioned on (not the full data set), their targets,
    kernel identifiers, hyper-parameters with the Gaussian noise, and the source's noise_diag."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_model_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        np.testing.assert_array_equal(ret["Z"], ret["z"])
        np.testing.assert_array_equal(ret["Y"], ret["y"])
        assert ret["Z"].shape == (5, 3) and ret["shapes"] == ((2, 5), (2, 128, 128))
        spec = ret["spec"]
        assert spec["kern_types"] == ["rbf", "mat52"] and spec["noise_diag"] == 3e-5
        assert (spec["n_s_out"], spec["n_s_in"], spec["n_u"]) == (2, 2, 1)
        np.testing.assert_array_equal(ret["noise"], [0.02, 0.03])
        np.testing.assert_array_equal(ret["ls1"], [1.5, 1.0, 0.5])


def _packed_worker(rank, world, port, N, n_out, piece, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # host stand-ins of sr_gp_export_packed / sr_gp_import_packed: row i contributes U[i, i:]
        rng = np.random.default_rng(5)
        U = np.triu(rng.standard_normal((n_out, N, N)))
        recv = np.zeros_like(U)
        calls = []

        def export_piece(d, r0, r1, buf):
            buf.copy_(torch.from_numpy(np.concatenate([U[d, i, i:] for i in range(r0, r1)])))

        def import_piece(d, r0, r1, buf):
            calls.append((d, r0, r1))
            flat, o = buf.numpy(), 0
            for i in range(r0, r1):
                recv[d, i, i:] = flat[o:o + N - i]
                o += N - i
            assert o == flat.size

        moved, k = parallel.broadcast_packed_factor(N, n_out, export_piece if rank == 0 else None,
                                                    import_piece if rank != 0 else None, src=0,
                                                    device=torch.device("cpu"), piece_doubles=piece)
        if rank == 1:
            ret["equal"] = bool(np.array_equal(recv, U))
            ret["moved"], ret["pieces"], ret["calls"] = moved, k, calls
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_packed_factor_broadcast_world2():
    """the chunked, double-buffered broadcast of the packed triangle: every piece arrives once, in order, and the
    receiver rebuilds exactly the sender's upper triangle; N (N + 1) / 2 doubles per output travel."""
    N, n_out, piece = 37, 2, 100
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_packed_worker, args=(2, _free_port(), N, n_out, piece, ret), nprocs=2, join=True)
        assert ret["equal"]
        assert ret["moved"] == n_out * N * (N + 1) // 2 * 8
        pcs = parallel.packed_pieces(N, piece)
        assert ret["pieces"] == n_out * len(pcs) and len(pcs) > 3
        assert ret["calls"] == [(d, r0, r1) for d in range(n_out) for (r0, r1, _) in pcs]


def test_packed_pieces_cover_the_triangle():
    for N in (1, 2, 127, 128, 5000, 50000):
        for cap in (1, 1000, 8 << 20):
            if N > 5000 and cap < 1000:
                continue
            pcs = parallel.packed_pieces(N, cap)
            assert pcs[0][0] == 0 and pcs[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(pcs[:-1], pcs[1:]))
            assert sum(c for _, _, c in pcs) == N * (N + 1) // 2
            # a piece only exceeds the cap when it is a single row
            assert all(c <= cap or r1 - r0 == 1 for r0, r1, c in pcs)


def test_shard_bounds_partition():
    for T in (0, 1, 7, 64, 65536, 8388608 + 3):
        for world in (1, 2, 3, 8):
            cuts = [parallel.shard_bounds(T, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == T
            for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
