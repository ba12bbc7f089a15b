#!/usr/bin/env python3
"""Batched counterpart of the reference's uncertainty-propagation experiment
(/root/reference/safe_exploration/uncertainty_propagation_runner.py:28-42): for each of n_rollouts random
affine control sequences (k_fb = .1 randn(n_safe-1, n_u, n_s), k_ff = .1 randn(n_safe, n_u)) propagate the
ellipsoidal over-approximation of the state n_safe steps through the GP dynamics model and report how many
trajectories stay inside the safety polytope |x_i| <= 1.  All rollouts run as ONE batch on the GPU.

    python examples/uncertainty_propagation_batch.py [n_rollouts] [n_safe] [N]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, gp_reachability as reach, workload  # noqa: E402


def main():
    n_rollouts = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_safe = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 150           # m=150 of defaultconfig_episode.py:34
    n_s, n_u = 4, 1                                               # cart-pole dims (environments.py:657)
    prob = workload.make_problem(0, N, n_s, n_u, 1, sf2=0.01)
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob))
    gp.update_model(prob["Z"], prob["Y"], opt_hyp=False, replace_old=True, choose_data=False)
    roll = workload.random_rollout_controls(1, n_rollouts, n_safe, n_s, n_u)
    l_mu = l_sigm = np.array([0.05] * n_s)                        # environments.py:702-704
    a, b = 0.9 * np.eye(n_s), 0.05 * np.ones((n_s, n_u))         # linear prior model (conf.lin_prior)
    t0 = time.time()
    p_all, q_all = reach.multistep_reachability_batch(roll["p0"], gp, roll["k_fb"], roll["k_ff"], l_mu, l_sigm,
                                                      None, 2.0, a, b)
    h_mat = np.vstack((np.eye(n_s), -np.eye(n_s)))
    h_vec = np.ones((2 * n_s, 1))
    d = reach.lin_ellipsoid_safety_distance_batch(p_all.reshape(-1, n_s), q_all.reshape(-1, n_s, n_s), h_mat,
                                                  h_vec, 2.0).reshape(n_rollouts, n_safe, -1)
    safe = np.all(d < 0, axis=(1, 2))
    dt = time.time() - t0
    print("%d rollouts x %d steps (N=%d): %.1f ms, %.0f step-evals/s; %d/%d trajectories certified safe; "
          "mean trace(Q) per step: %s" % (n_rollouts, n_safe, N, 1e3 * dt, n_rollouts * n_safe / dt, safe.sum(),
                                          n_rollouts, np.round(np.trace(q_all, axis1=2, axis2=3).mean(0), 4)))


if __name__ == "__main__":
    main()
