"""ctypes binding of include/safereach.h (libsafereach.so, gfx950).

There is no CPU fallback: if the shared library is missing this module raises at import, and every
compute entry point raises RuntimeError when the HIP runtime reports an error (e.g. no device).
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  MUST precede CDLL below: PyTorch-ROCm bundles its own libamdhip64.so.7;
#               loading it first makes libsafereach.so bind to that same runtime (one HIP runtime
#               per process -- two of them cannot both own the device).

from ._build import LIB_PATH

SR_OK, SR_EINVAL, SR_EHIP, SR_ENOTPD, SR_ESTATE, SR_EUNSUPPORTED, SR_EBUSY = 0, -1, -2, -3, -4, -5, -6
K_GRAM, K_POTRF, K_GEMM, K_KSTAR, K_VAR, K_FINAL, K_ELL, K_TRINV, K_SMALL = range(9)
KERNEL_NAMES = {K_GRAM: "sr_gram_kernel", K_POTRF: "sr_potrf_diag_kernel", K_GEMM: "sr_gemm_tn_kernel",
                K_KSTAR: "sr_kstar_kernel", K_VAR: "sr_var_kernel", K_FINAL: "sr_finalize_kernel",
                K_ELL: "sr_ellipsoid_kernel"}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libsafereach.so not found at %s -- build it with `make -C safe_exploration_amd/csrc` or `python __graft_entry__.py` "
        "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_P = ctypes.c_void_p          # device pointers / stream
_I = ctypes.c_int
_L = ctypes.c_long
_D = ctypes.c_double
_PI = ctypes.POINTER(ctypes.c_int)
_PL = ctypes.POINTER(ctypes.c_long)
_PD = ctypes.POINTER(ctypes.c_double)
_H = ctypes.c_void_p          # sr_gp_t

# name -> (restype, argtypes); mirrors include/safereach.h one to one
SIGNATURES = {
    "sr_version": (_I, []),
    "sr_last_error": (ctypes.c_char_p, []),
    "sr_device_count": (_I, [_PI]),
    "sr_gp_create": (_I, [ctypes.POINTER(_H), _I, _I, _I, _I]),
    "sr_gp_destroy": (_I, [_H]),
    "sr_gp_set_data": (_I, [_H, _P, _P, _P, _P, _P, _P]),
    "sr_gp_set_data_general": (_I, [_H, _P, _P, _P, _P, _P]),
    "sr_gp_factorize": (_I, [_H, _P, _PI]),
    "sr_gp_append": (_I, [_H, _P, _P, _I, _P, _PI]),
    "sr_gp_append1_host": (_I, [_H, _P, _P, _P, _PI]),
    "sr_gp_padded_n": (_I, [_H, _PL]),
    "sr_gp_dims": (_I, [_H, _PI, _PI, _PI, _PL]),
    "sr_gp_export": (_I, [_H, _P, _P, _P]),
    "sr_gp_import": (_I, [_H, _P, _P, _P]),
    "sr_gp_packed_count": (_L, [_H, _L, _L]),
    "sr_gp_export_packed": (_I, [_H, _I, _L, _L, _P, _P]),
    "sr_gp_import_begin": (_I, [_H, _P, _P]),
    "sr_gp_import_packed": (_I, [_H, _I, _L, _L, _P, _P]),
    "sr_gp_import_end": (_I, [_H]),
    "sr_gp_inv_k": (_I, [_H, _I, _P, _P]),
    "sr_gp_predict": (_I, [_H, _P, _L, _P, _P, _P, _P]),
    "sr_gp_linearize": (_I, [_H, _P, _P, _P, _P, _P, _P, _P]),
    "sr_gp_set_input_transform": (_I, [_H, _P, _I, _P]),
    "sr_onestep_reach": (_I, [_H, _L, _P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P, _P]),
    "sr_multistep_reach": (_I, [_H, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _P]),
    "sr_multistep_moments": (_I, [_H, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sr_moment_step": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sr_ellipsoid_step": (_I, [_I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P,
                               _P, _P]),
    "sr_remainder_overapprox": (_I, [_I, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sr_safety_distance": (_I, [_I, _L, _I, _I, _P, _P, _P, _P, _D, _P, _P]),
    "sr_distance_to_center": (_I, [_I, _L, _I, _I, _P, _I, _P, _P, _P, _P]),
    "sr_gp_mll": (_I, [_H, _P, _P, _P]),
    "sr_gp_logdet": (_I, [_H, _P, _P]),
    "sr_gp_logdet_cached": (_I, [_H, _P]),
    "sr_gp_sample": (_I, [_I, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sr_gp_set_chunk": (_I, [_H, _L]),
    "sr_gp_set_var_group": (_I, [_H, _I]),
    "sr_gp_set_var_variant": (_I, [_H, _I]),
    "sr_gp_set_chain": (_I, [_H, _I]),
    "sr_gp_release_scratch": (_I, [_H]),
    "sr_release_cached_memory": (_I, []),
    "sr_publish": (_I, [_I, _P, _I, _P, _P, ctypes.c_ulonglong, _P]),
    "sr_wait_flag": (_I, [_P, ctypes.c_ulonglong, ctypes.c_double]),
    "sr_host_block_is_device_visible": (_I, [_I, _P]),
    "sr_stream_synchronize": (_I, [_I, _P]),
    "sr_gp_call1": (_I, [_H, _P, _I, _P, _P, ctypes.c_ulonglong, _P]),
    "sr_gp_server_start": (_I, [_H, ctypes.c_double]),
    "sr_gp_server_stop": (_I, [_H]),
    "sr_gp_server_call": (_I, [_H, _P, _I, _P, ctypes.c_double]),
    "sr_gp_server_state": (_I, [_H, _P, _P, _P, _P]),
    "sr_gp_last_chain": (_I, [_H]),
    "sr_gp_chain_status": (_I, [_H, _PI]),
    "sr_gp_set_small_path": (_I, [_H, _I]),
    "sr_gp_set_fact_panel": (_I, [_H, _I]),
    "sr_gp_set_fact_pipeline": (_I, [_H, _I]),
    "sr_gp_fact_pipelined": (_I, [_H]),
    "sr_gp_flow_stats": (_I, [_H, ctypes.POINTER(ctypes.c_uint), _I]),
    "sr_test_flow_plan": (_I, [_I, _I, _I, _PI, _PL]),
    "sr_test_flow_fail": (_I, [_I]),
    "sr_test_gemm_tn": (_I, [_I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _D, _D, _I, _P]),
    "sr_test_gemm_tn_upper": (_I, [_I, _P, _L, _P, _L, _P, _L, _I, _I, _I, _D, _D, _I, _P]),
    "sr_test_potrf_diag": (_I, [_I, _P, _L, _P, _P, _L, _P, _I, _P]),
    "sr_test_chain_drop": (_I, [_H, _I]),
    "sr_test_grid_append_abort": (_I, [_I]),
    "sr_gp_grid_append_aborts": (_I, [_H, _PL]),
    "sr_prof_enable": (_I, [_H, _I]),
    "sr_prof_reset": (_I, [_H]),
    "sr_prof_get": (_I, [_H, _I, _PD, _PL]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)        # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.sr_last_error().decode("utf-8", "replace")


def check(rc):
    """Translate a C-ABI status into the exception the reference surface would raise."""
    if rc == SR_OK:
        return
    msg = "libsafereach: %s (code %d)" % (last_error(), rc)
    if rc == SR_ENOTPD:
        raise np.linalg.LinAlgError(msg)
    if rc == SR_EINVAL:
        raise ValueError(msg)
    if rc == SR_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def device_count():
    n = ctypes.c_int(0)
    rc = lib.sr_device_count(ctypes.byref(n))
    return n.value if rc == SR_OK else 0
