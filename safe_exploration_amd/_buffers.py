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


def as_dev(x, device, shape=None):
    """float64, contiguous tensor on `device` (no copy when it already is one)."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.float64)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64))).to(device)
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
    return torch.empty(shape, dtype=torch.float64, device=device)


def zeros_i32(n, device):
    return torch.zeros(n, dtype=torch.int32, device=device)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def to_numpy(t):
    return t.detach().cpu().numpy()
