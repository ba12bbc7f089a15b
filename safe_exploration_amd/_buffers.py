"""PyTorch-ROCm tensors as the device-buffer carrier for the C-ABI (plumbing only)."""
import ctypes

import numpy as np
import torch


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("safe_exploration_amd needs a ROCm GPU (gfx950): there is no CPU fallback")


def resolve_device(device=None):
    require_gpu()
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise ValueError("device must be a cuda/ROCm device, got %s" % (device,))
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def _retry_after_release(fn):
    """Run ``fn``; if torch runs out of device memory, hand the blocks libsafereach keeps for re-use back to the driver
    (sr_release_cached_memory: torch's allocator cannot reclaim them) and try once more."""
    try:
        return fn()
    except torch.cuda.OutOfMemoryError:
        from . import _lib
        _lib.lib.sr_release_cached_memory()
        torch.cuda.empty_cache()
        return fn()


def as_dev(x, device, shape=None):
    """float64, contiguous tensor on `device` (no copy when it already is one)."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        t = _retry_after_release(lambda: x.to(device=device, dtype=torch.float64))
    else:
        h = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
        t = _retry_after_release(lambda: h.to(device))
    if shape is not None:
        t = t.reshape(shape)
    return t.contiguous()


_CONST_CACHE = {}


def const_dev(x, device, shape):
    """Small constant host array (model matrices a/b, Lipschitz vectors) -> device tensor, memoised on
    its bytes so that repeated calls of the per-step entry points do not pay an H2D copy each."""
    if isinstance(x, torch.Tensor):
        return as_dev(x, device, shape)
    arr = np.ascontiguousarray(np.asarray(x, dtype=np.float64)).reshape(shape)
    key = (arr.tobytes(), tuple(shape), str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 256:
            _CONST_CACHE.clear()
        t = torch.from_numpy(arr.copy()).to(device)
        _CONST_CACHE[key] = t
    return t


def empty(shape, device):
    return _retry_after_release(lambda: torch.empty(shape, dtype=torch.float64, device=device))


def zeros_i32(n, device):
    return torch.zeros(n, dtype=torch.int32, device=device)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class RawStream(object):
    """The current stream of a device as its raw handle (``cuda_stream``) with ``synchronize()``: what the entry points
    need of ``torch.cuda.current_stream(device)``, whose Stream object costs 2 - 3 us to build per call."""
    __slots__ = ("cuda_stream", "index")

    def __init__(self, device):
        if _raw_stream is not None and device.index is not None:
            self.index = device.index
            self.cuda_stream = _raw_stream(device.index)
        else:
            st = torch.cuda.current_stream(device)
            self.index = st.device.index
            self.cuda_stream = st.cuda_stream

    def synchronize(self):
        # (with the stream's device current: handle 0 -- PyTorch's default stream -- is the null stream of whichever
        #  device is current, and a model may live on another one than the caller's)
        from . import _lib
        _lib.check(_lib.lib.sr_stream_synchronize(self.index, ctypes.c_void_p(self.cuda_stream)))


def current_stream(device):
    return RawStream(device)


def stream_ptr(device):
    return ctypes.c_void_p(RawStream(device).cuda_stream)


def to_numpy(t):
    return t.detach().cpu().numpy()


STAGING_MAX_DOUBLES = 1 << 22       # 32 MB of pinned memory at most per direction; bigger calls copy argument by argument
import os as _os
# inputs + outputs of a call up to this many doubles are not copied at all (Staging.stage); 0 switches it off
ZERO_COPY_DOUBLES = int(_os.environ.get("SR_ZERO_COPY_DOUBLES", "512"))


class Staging(object):
    """Packed host <-> device staging for an entry point called with NumPy arrays: ONE pinned block in (all inputs back
    to back, one asynchronous H2D copy), ONE packed device block out and one pinned block back (one D2H copy, one stream
    synchronisation).  Copying every argument on its own from pageable memory and every result back on its own cost a
    256-rollout, 15-step chain 105 us of host time on top of the kernel's 105 us.  Grow-only; kept with the model handle."""

    def __init__(self, device):
        self.device = device
        self.h_in = self.d_in = self.h_out = self.d_out = None
        self._plans, self._gen, self._last = {}, 0, None
        self._vis, self._vis_gen = False, -1

    def _room(self, n_in, n_out):
        if self.h_in is None or self.h_in.numel() < n_in:
            # (room to grow: a caller whose batches get a little longer every call must not pin a new block each time)
            n_in = max(n_in, 1024, 0 if self.h_in is None else min(2 * self.h_in.numel(), STAGING_MAX_DOUBLES))
            self.h_in = torch.empty(n_in, dtype=torch.float64).pin_memory()
            self.d_in = torch.empty(self.h_in.numel(), dtype=torch.float64, device=self.device)
            self.h_in_np = self.h_in.numpy()
            self._gen += 1
        if self.h_out is None or self.h_out.numel() < n_out:
            n_out = max(n_out, 1024, 0 if self.h_out is None else min(2 * self.h_out.numel(), STAGING_MAX_DOUBLES))
            self.h_out = torch.empty(n_out, dtype=torch.float64).pin_memory()
            self.d_out = torch.empty(self.h_out.numel(), dtype=torch.float64, device=self.device)
            self.h_out_np = self.h_out.numpy()
            self._gen += 1

    def _visible(self):
        """The pinned blocks are device-visible at their own addresses (asked once per pair of blocks)."""
        if self._vis_gen != self._gen:
            from . import _lib
            dev = self.device.index or 0
            self._vis = all(_lib.lib.sr_host_block_is_device_visible(dev, ctypes.c_void_p(t.data_ptr())) == 1
                            for t in (self.h_in, self.h_out))
            self._vis_gen = self._gen
        return self._vis

    def stage(self, arrays, out_shapes, zero_copy=False):
        """arrays: NumPy float64 arrays (or None); out_shapes: shapes of the results.  Returns (device views of the
        inputs -- None where the input was None --, device views of the outputs).  The views of a call signature
        (shapes) are built once: slicing tensors costs microseconds each, and a caller repeats its shapes.
        zero_copy: the caller's entry point hands these pointers to KERNELS only (no copy command reads or writes them), so
        a handful of numbers may stay in the pinned blocks."""
        key = (tuple(None if a is None else a.shape for a in arrays), tuple(tuple(sh) for sh in out_shapes), bool(zero_copy))
        plan = self._plans.get(key)
        if plan is None or plan["gen"] != self._gen:
            n_in = sum(int(a.size) for a in arrays if a is not None)
            n_out = sum(int(np.prod(sh)) for sh in out_shapes)
            self._room(n_in, n_out)
            if len(self._plans) > 64:
                self._plans.clear()
            # a handful of numbers (one query, one rollout): the kernels read the pinned block itself and write their
            # results into the pinned result block -- pinned host memory is device-visible at its own address --, so the
            # call costs no copy command in either direction (ZERO_COPY_DOUBLES; one-step reachability of one query, NumPy
            # in and out: 50 -> 40 us)
            direct = zero_copy and ZERO_COPY_DOUBLES > 0 and n_in + n_out <= ZERO_COPY_DOUBLES and self._visible()
            src_in, src_out = (self.h_in, self.h_out) if direct else (self.d_in, self.d_out)
            views, hviews, off = [], [], 0
            for a in arrays:
                if a is None:
                    views.append(None)
                    hviews.append(None)
                    continue
                n = int(a.size)
                hviews.append(self.h_in_np[off:off + n].reshape(a.shape))
                views.append(src_in[off:off + n].view(a.shape))
                off += n
            outs, houts, ooff = [], [], 0
            for sh in out_shapes:
                n = int(np.prod(sh))
                outs.append(src_out[ooff:ooff + n].view(tuple(sh)))
                houts.append(self.h_out_np[ooff:ooff + n].reshape(tuple(sh)))
                ooff += n
            plan = {"gen": self._gen, "views": views, "hviews": hviews, "outs": outs, "houts": houts, "direct": direct,
                    "d_in": self.d_in[:off] if off else None, "h_in": self.h_in[:off] if off else None,
                    "d_out": self.d_out[:ooff], "h_out": self.h_out[:ooff]}
            self._plans[key] = plan
        for a, hv in zip(arrays, plan["hviews"]):
            if a is not None:
                np.copyto(hv, a)
        if plan["d_in"] is not None and not plan["direct"]:
            plan["d_in"].copy_(plan["h_in"], non_blocking=True)
        self._last = plan
        return plan["views"], plan["outs"]

    def fetch(self):
        """The outputs of the last ``stage`` as NumPy arrays (copies); synchronises the current stream."""
        plan = self._last
        if not plan["direct"]:
            plan["h_out"].copy_(plan["d_out"], non_blocking=True)
        RawStream(self.device).synchronize()
        return [h.copy() for h in plan["houts"]]
