"""safe_exploration_amd -- MI355X-native GP-dynamics inference + ellipsoid reachability.

Drop-in for the hot path of befelix/safe-exploration (and nothing else):

    from safe_exploration_amd import SimpleGPModel, gp_reachability, utils_ellipsoid, utils

Importing the package loads libsafereach.so (built in-tree by ``make -C safe_exploration_amd/csrc``); it fails loudly when the library is missing, and every compute
call fails loudly when no ROCm GPU is visible -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (ImportError here == library not built)
from .state_space_models import StateSpaceModel  # noqa: F401
from .ssm_hip.gaussian_process import SimpleGPModel  # noqa: F401
from . import gp_reachability, utils, utils_ellipsoid  # noqa: F401



def release_cached_memory():
    """Hand the device buffers the library keeps for re-use (sr_release_cached_memory) back to the driver."""
    _lib.check(_lib.lib.sr_release_cached_memory())


__all__ = ["SimpleGPModel", "StateSpaceModel", "gp_reachability", "utils", "utils_ellipsoid", "release_cached_memory"]
