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


def model_spec(gp):
    """Everything but the arrays that a receiver needs to rebuild ``gp``: dimensions, kernel identifiers, the
    hyper-parameters INCLUDING the Gaussian noise (``hyp`` alone does not carry it) and the ``noise_diag`` the
    source was trained with."""
    hyp = []
    for i, h in enumerate(gp.hyp):
        h = {k: np.asarray(v, dtype=np.float64).tolist() for k, v in h.items()}
        h["noise_variance"] = float(gp._noise[i])
        hyp.append(h)
    return {"n_s_out": gp.n_s_out, "n_s_in": gp.n_s_in, "n_u": gp.n_u, "kern_types": list(gp.kern_types),
            "hyp": hyp, "noise_diag": float(gp._noise_diag if gp._noise_diag is not None else 1e-5)}


def broadcast_model_state(gp, src=0, device=None):
    """The communication half of ``replicate_model``: returns (spec, tensors) on every rank -- the model
    description as a Python object and Z / targets / alpha / U^-1 as tensors on ``device``.  Only rank ``src``
    reads ``gp``.  Backend-agnostic (RCCL on the GPUs, gloo in the CPU tests)."""
    rank = dist.get_rank()
    spec = [model_spec(gp) if rank == src else None]
    dist.broadcast_object_list(spec, src=src)
    payload = None
    if rank == src:
        alpha, wt = gp.export_state()
        payload = {"Z": torch.from_numpy(np.ascontiguousarray(gp.z_fit)),
                   "Y": torch.from_numpy(np.ascontiguousarray(gp.y_z)),
                   "alpha": alpha, "wt": wt}
    return spec[0], broadcast_tensors(payload, src=src, device=device)


PIECE_DOUBLES = 8 << 20       # staging piece of the packed factor: 64 MB (two of them live on every rank)


def packed_pieces(N, max_doubles=PIECE_DOUBLES):
    """Row ranges [(row0, row1, count)] of the packed upper triangle (row i carries N - i doubles), each at most
    ``max_doubles`` long (a single row may exceed it)."""
    out, r0, N = [], 0, int(N)
    while r0 < N:
        # largest r1 with (r1 - r0) N - (r1 (r1 - 1) - r0 (r0 - 1)) / 2 <= max_doubles
        lo, hi = r0 + 1, N
        cnt = lambda r1: (r1 - r0) * N - (r1 * (r1 - 1) - r0 * (r0 - 1)) // 2
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if cnt(mid) <= max_doubles:
                lo = mid
            else:
                hi = mid - 1
        out.append((r0, lo, cnt(lo)))
        r0 = lo
    return out


LAST_REPLICATION = {}         # statistics of the last replicate_model on this rank (bytes moved, pieces)


def broadcast_packed_factor(N, n_out, export_piece, import_piece, src=0, device=None, piece_doubles=PIECE_DOUBLES):
    """The packed upper triangle of U^-1 of every output from rank ``src`` to all others, in pieces of at most
    ``piece_doubles`` through TWO staging buffers: while piece k travels, piece k+1 is packed on the sender and piece
    k-1 unpacked on the receivers.  ``export_piece(d, row0, row1, buf)`` fills ``buf`` on the sender,
    ``import_piece(d, row0, row1, buf)`` consumes it on a receiver (either may be None on the ranks that do not need
    it).  Returns (bytes moved, pieces)."""
    rank = dist.get_rank()
    pieces = packed_pieces(N, piece_doubles)
    cap = max(c for _, _, c in pieces)
    stage = [torch.empty(cap, dtype=torch.float64, device=device) for _ in range(2)]
    inflight = [None, None]

    def retire(b):
        if inflight[b] is not None:
            work, args = inflight[b]
            work.wait()
            if rank != src:
                import_piece(*args)
            inflight[b] = None

    moved = k = 0
    for d in range(n_out):
        for (r0, r1, cnt) in pieces:
            b = k & 1
            retire(b)                              # the broadcast that last used this staging buffer
            buf = stage[b][:cnt]
            if rank == src:
                export_piece(d, r0, r1, buf)
            inflight[b] = (dist.broadcast(buf, src=src, async_op=True), (d, r0, r1, buf))
            moved += cnt * 8
            k += 1
    retire(k & 1)
    retire((k + 1) & 1)
    return moved, k


def replicate_model(gp, prob=None, src=0, device=None, piece_doubles=PIECE_DOUBLES):
    """Rank `src` holds a trained HIP SimpleGPModel; every other rank receives the model description (object
    broadcast: dimensions, kernels, hyper-parameters, noise), then Z, the targets that belong to Z and alpha, then
    the PACKED upper triangle of U^-1 -- N (N + 1) / 2 doubles per output instead of the Np^2 of the dense buffer, in
    pieces of <= 64 MB through two staging buffers (no copy of the factor's size on either side) -- and adopts it
    without factorising (sr_gp_import_begin / _packed / _end).  RCCL over xGMI with backend "nccl", gloo in the tests.
    ``prob`` is accepted for backward compatibility and ignored: the source model is the single source of truth.
    Returns the rank-local model."""
    from .ssm_hip.gaussian_process import SimpleGPModel
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    rank = dist.get_rank()
    spec = [model_spec(gp) if rank == src else None]
    dist.broadcast_object_list(spec, src=src)
    spec = spec[0]
    payload = None
    if rank == src:
        payload = {"Z": torch.from_numpy(np.ascontiguousarray(gp.z_fit)),
                   "Y": torch.from_numpy(np.ascontiguousarray(gp.y_z)), "alpha": gp.export_alpha()}
    got = broadcast_tensors(payload, src=src, device=dev)
    N, n_out = got["Z"].shape[0], spec["n_s_out"]
    if rank == src:
        local = gp
    else:
        local = SimpleGPModel(spec["n_s_out"], spec["n_s_in"], spec["n_u"], kern_types=spec["kern_types"],
                              hyp=spec["hyp"], device=dev)
        local.begin_import(got["Z"].cpu().numpy(), got["Y"].cpu().numpy(), got["alpha"],
                           noise_diag=spec["noise_diag"])
    moved, k = broadcast_packed_factor(N, n_out, local.export_packed if rank == src else None,
                                       local.import_packed if rank != src else None, src=src, device=dev,
                                       piece_doubles=piece_doubles)
    if rank != src:
        local.end_import()
    torch.cuda.current_stream(dev).synchronize()
    LAST_REPLICATION.clear()
    LAST_REPLICATION.update({"factor_bytes": moved, "other_bytes": sum(int(t.numel()) * 8 for t in got.values()),
                             "pieces": k, "dense_factor_bytes": n_out * (-(-N // 128) * 128) ** 2 * 8})
    return local


def fit_outputs_sharded(n_s_out, n_s_in, n_u, Z, Y, kern_types=None, hyp=None, noise_diag=1e-5, device=None):
    """Model update with the OUTPUTS sharded over the ranks (the large-N alternative): the n_s_out
    Gaussian processes are independent problems, so rank r factorises outputs d = r, r + world, ... on its GPU with
    no communication, then every factor is broadcast once from its owner (RCCL over xGMI; alpha and the packed upper
    triangle of U^-1) and each rank adopts the complete posterior (``begin_import`` / ``import_packed`` / ``end_import``).  Returns the rank-local full model.
    Config 4 (N = 50000, n_out = 2) on two GPUs: half the factorisation time plus one 20 GB broadcast."""
    from .ssm_hip.gaussian_process import SimpleGPModel
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    kern_types = list(kern_types) if kern_types is not None else ["rbf"] * n_s_out
    hyp = list(hyp) if hyp is not None else [None] * n_s_out
    Z = np.asarray(Z, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    mine = [d for d in range(n_s_out) if d % world == rank]
    sub = None
    if mine:
        sub = SimpleGPModel(len(mine), n_s_in, n_u, kern_types=[kern_types[d] for d in mine],
                            hyp=[hyp[d] for d in mine], device=dev)
        sub.train(Z, Y[:, mine], opt_hyp=False, noise_diag=noise_diag)
    N = Z.shape[0]
    # alpha of every output from its owner (n_out x N doubles), then the factors: PACKED upper triangles, one output
    # after the other from its owner, in pieces of <= 64 MB through two staging buffers (no dense Np x Np copy anywhere:
    # 20 instead of 40 GB on the wire for config 4, and no second resident copy of the factor on any rank)
    alpha = torch.empty((n_s_out, N), dtype=torch.float64, device=dev)
    mine_alpha = sub.export_alpha() if mine else None
    for d in range(n_s_out):
        owner = d % world
        if owner == rank:
            alpha[d].copy_(mine_alpha[mine.index(d)])
        dist.broadcast(alpha[d], src=owner)
    full = SimpleGPModel(n_s_out, n_s_in, n_u, kern_types=kern_types, hyp=hyp, device=dev)
    full.begin_import(Z, Y, alpha, noise_diag=noise_diag)
    moved = 0
    for d in range(n_s_out):
        owner = d % world

        def export_piece(_d, r0, r1, buf, d=d):
            sub.export_packed(mine.index(d), r0, r1, buf)
            full.import_packed(d, r0, r1, buf)            # the owner adopts its own rows from the same staging buffer

        def import_piece(_d, r0, r1, buf, d=d):
            full.import_packed(d, r0, r1, buf)

        m, _ = broadcast_packed_factor(N, 1, export_piece if owner == rank else None,
                                       import_piece if owner != rank else None, src=owner, device=dev)
        moved += m
    full.end_import()
    LAST_REPLICATION.clear()
    LAST_REPLICATION.update({"factor_bytes": moved, "other_bytes": int(alpha.numel()) * 8, "pieces": None,
                             "dense_factor_bytes": n_s_out * (-(-N // 128) * 128) ** 2 * 8})
    return full


def gather_rows(local, dst=0):
    """Gather per-rank row blocks (numpy, possibly different lengths) on `dst` in rank order."""
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, objs, dst=dst)
    if dist.get_rank() != dst:
        return None
    return np.concatenate(objs, axis=0)
