#!/usr/bin/env python3
"""Randomised sequences of model operations around the in-place one-point append (sr_gp::slide): single points, several
points at once, single queries, big batches (tile kernels: plain buffers first), linearisations, refits of the same data,
information gain -- in random order on models of 520 .. 1400 points, 1 .. 4 outputs, RBF and the journal kernels; after every
few operations the model is compared with one fitted on the same data from scratch.  GPU box:  python scripts/fuzz_append.py [seeds] [ops]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from safe_exploration_amd import SimpleGPModel, workload
from call_latency import kern_hyp

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nops = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = {"mu": 0.0, "var": 0.0, "ig": 0.0}
counts = {}
t_start = time.time()
for seed in range(seeds):
    rng = np.random.default_rng(1000 + seed)
    n_s = int(rng.integers(1, 5)); n_u = 1; D = n_s + n_u
    kt = ["rbf", "lin_mat52", "mat52", "lin_rbf"][int(rng.integers(0, 4))]
    N0 = int(rng.integers(520, 1100))
    prob = workload.make_problem(300 + seed, N0 + 400, n_s, n_u, 8)
    Z, Y = prob["Z"], prob["Y"]
    hyp = workload.hyp_list(prob) if kt == "rbf" else kern_hyp(kt, np.random.default_rng(seed), D, n_s)
    mk = lambda: SimpleGPModel(n_s, n_s, n_u, kern_types=[kt] * n_s, hyp=hyp, device="cuda:0")
    gp = mk(); gp.append_limit = 10 ** 9
    gp.train(Z[:N0], Y[:N0], opt_hyp=False)
    n = N0
    xq = np.hstack((rng.uniform(-1, 1, (2500, n_s)), rng.uniform(-1, 1, (2500, n_u))))
    for op_i in range(nops):
        op = rng.choice(["add1", "add1", "add1", "add1", "addm", "q1", "qbig", "lin", "refit", "ig"])
        counts[op] = counts.get(op, 0) + 1
        if op == "add1" and n + 1 <= Z.shape[0]:
            gp.update_model(Z[n:n + 1], Y[n:n + 1], opt_hyp=False, replace_old=False); n += 1
        elif op == "addm":
            m = int(rng.integers(2, 24))
            if n + m <= Z.shape[0]:
                gp.update_model(Z[n:n + m], Y[n:n + m], opt_hyp=False, replace_old=False); n += m
        elif op == "q1":
            gp.predict(xq[:1])
        elif op == "qbig":
            gp.predict(xq[:int(rng.choice([300, 1200, 2500]))])
        elif op == "lin":
            gp.linearize_predict(xq[:1, :n_s], xq[:1, n_s:], True)
        elif op == "refit":
            gp.train(Z[:n], Y[:n], opt_hyp=False)
        elif op == "ig":
            gp.information_gain()
        if op_i % 7 == 6 or op_i == nops - 1:
            ref = mk(); ref.train(Z[:n], Y[:n], opt_hyp=False)
            m1, v1 = gp.predict(xq[:64]); m2, v2 = ref.predict(xq[:64])
            e_mu = float(np.abs(m1 - m2).max() / max(1.0, np.abs(m2).max())); e_var = float(np.abs(v1 - v2).max())
            e_ig = float(np.abs(np.asarray(gp.information_gain()) - np.asarray(ref.information_gain())).max())
            worst["mu"] = max(worst["mu"], e_mu); worst["var"] = max(worst["var"], e_var); worst["ig"] = max(worst["ig"], e_ig)
            assert e_mu < 1e-8 and e_var < 1e-9 and e_ig < 1e-6, (seed, op_i, op, n, kt, n_s, e_mu, e_var, e_ig)
            del ref
    print("seed %d: %-9s n_out=%d  N %d -> %d  ok" % (seed, kt, n_s, N0, n), flush=True)
    del gp
print("operations: %s" % ", ".join("%s %d" % kv for kv in sorted(counts.items())))
print("worst deviation from a model fitted from scratch: mean %.2e (relative), variance %.2e, information gain %.2e; %.0f s" % (
    worst["mu"], worst["var"], worst["ig"], time.time() - t_start))
