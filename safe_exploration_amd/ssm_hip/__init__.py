"""HIP-backed GP state-space model (counterpart of the reference's ssm_gpy sub-package)."""
from .gaussian_process import SimpleGPModel  # noqa: F401
