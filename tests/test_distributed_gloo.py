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


def test_shard_bounds_partition():
    for T in (0, 1, 7, 64, 65536, 8388608 + 3):
        for world in (1, 2, 3, 8):
            cuts = [parallel.shard_bounds(T, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == T
            for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
