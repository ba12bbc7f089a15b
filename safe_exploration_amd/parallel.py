"""Multi-GPU: shard the query batch, replicate the model once.

The path is embarrassingly parallel over query states: one process per GPU, each owns
a contiguous slice of the batch; the only communication is a ONE-TIME broadcast of the training
set and the cached posterior state (alpha, U^-1) from the rank that factorised -- RCCL over xGMI via
``torch.distributed`` (backend "nccl"), gloo in the CPU tests.  No per-batch collective.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(T, world_size, rank):
    """Contiguous [lo, hi) slice of T items for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank {} outside world of {}".format(rank, world_size))
    base, rem = divmod(int(T), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_tensors(tensors, src=0, device=None):
    """Broadcast a dict name -> tensor from `src`.  Non-src ranks pass None (or anything) and receive
    freshly allocated tensors on `device`.  Shapes/dtypes travel first as a Python object."""
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in tensors.items()}
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for name, (shape, dtype) in meta[0].items():
        if rank == src:
            t = tensors[name].contiguous()
            if device is not None:
                t = t.to(device)
        else:
            t = torch.empty(shape, dtype=getattr(torch, dtype), device=device)
        dist.broadcast(t, src=src)
        out[name] = t
    return out


def replicate_model(gp, prob, src=0):
    """Rank `src` holds a trained HIP SimpleGPModel; every other rank receives Z, Y, alpha and U^-1
    over RCCL and adopts them without factorising.  `prob` supplies the (replicated, tiny)
    hyper-parameters on every rank.  Returns the rank-local model."""
    from .ssm_hip.gaussian_process import SimpleGPModel
    from .workload import hyp_list
    rank = dist.get_rank()
    n_s = len(prob["signal_var"])
    D = prob["lengthscale"].shape[1]
    dev = torch.device("cuda", torch.cuda.current_device())
    payload = None
    if rank == src:
        alpha, wt = gp.export_state()
        payload = {"Z": torch.from_numpy(np.ascontiguousarray(gp.z)).to(dev),
                   "Y": torch.from_numpy(np.ascontiguousarray(gp.y_train)).to(dev),
                   "alpha": alpha, "wt": wt}
    got = broadcast_tensors(payload, src=src, device=dev)
    if rank == src:
        return gp
    local = SimpleGPModel(n_s, n_s, D - n_s, kern_types=prob.get("kern_types", ["rbf"] * n_s),
                          hyp=prob.get("hyp", None) or hyp_list(prob), device=dev)
    local.import_state(got["Z"].cpu().numpy(), got["Y"].cpu().numpy(), got["alpha"], got["wt"])
    return local


def gather_rows(local, dst=0):
    """Gather per-rank row blocks (numpy, possibly different lengths) on `dst` in rank order."""
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, objs, dst=dst)
    if dist.get_rank() != dst:
        return None
    return np.concatenate(objs, axis=0)
