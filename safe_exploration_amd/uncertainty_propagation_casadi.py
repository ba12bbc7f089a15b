"""Name-compatible alias of ``uncertainty_propagation`` (the reference module is
``uncertainty_propagation_casadi.py``; here the same four functions are numeric batched kernels, nothing symbolic)."""
from .uncertainty_propagation import (one_step_taylor, multi_step_taylor_symbolic, mean_equivalent_multistep,  # noqa: F401
                                      one_step_mean_equivalent, multi_step_taylor, multistep_moments_batch,
                                      moment_step_batch, TAYLOR, MEAN_EQUIVALENT)
