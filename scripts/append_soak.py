#!/usr/bin/env python3
"""Soak test of the one-point append: hundreds of consecutive update_model(+1) calls across several padded sizes and both
one-launch routes (one workgroup per output up to 512 padded rows, the grid kernel beyond), mixed with predictions, then the
model against a refit on all the data.  GPU box:  python scripts/append_soak.py [N0] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, workload
from call_latency import kern_hyp
N0 = int(sys.argv[1]) if len(sys.argv) > 1 else 400
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 900
for kt, n_s in (("rbf", 2), ("lin_mat52", 4)):
    prob = workload.make_problem(21, N0 + steps, n_s, 1, 8)
    D = n_s + 1
    hyp = workload.hyp_list(prob) if kt == "rbf" else kern_hyp(kt, np.random.default_rng(3), D, n_s)
    gp = SimpleGPModel(n_s, n_s, 1, kern_types=[kt] * n_s, hyp=hyp, device="cuda:0")
    gp.append_limit = 10 ** 9
    Z, Y = prob["Z"], prob["Y"]
    gp.train(Z[:N0], Y[:N0], opt_hyp=False)
    t0 = time.perf_counter()
    for i in range(N0, N0 + steps):
        gp.update_model(Z[i:i + 1], Y[i:i + 1], opt_hyp=False, replace_old=False)
        if i % 97 == 0:
            gp.predict(Z[i - 3:i + 1])
    dt = (time.perf_counter() - t0) / steps * 1e6
    ig = np.asarray(gp.information_gain())
    ref = SimpleGPModel(n_s, n_s, 1, kern_types=[kt] * n_s, hyp=hyp, device="cuda:0")
    ref.train(Z[:N0 + steps], Y[:N0 + steps], opt_hyp=False)
    xq = np.vstack((Z[N0 + steps - 5:N0 + steps], Z[:5] + 0.01))
    m1, v1 = gp.predict(xq); m2, v2 = ref.predict(xq)
    print("%-9s n_out=%d  N %d -> %d: %.1f us per append; against the refit: max|mu| %.2e  max|var| %.2e  |info gain| %.2e" % (
        kt, n_s, N0, N0 + steps, dt, np.abs(m1 - m2).max(), np.abs(v1 - v2).max(), np.abs(ig - np.asarray(ref.information_gain())).max()), flush=True)
