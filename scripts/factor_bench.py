#!/usr/bin/env python3
"""Model-update timing: wall time of a warm refit (sr_gp_set_data + sr_gp_factorize) per size and panel width,
with the per-kernel split of the library's own hipEvent pairs and the posterior identity as a sanity check.

usage (GPU box):  python scripts/factor_bench.py [N ...]      (default 1000 2000 5000 10000)
SR_PANELS=0,2,4 picks the panel widths (0 = by size), SR_PIPE=0,1,2,3 the chain forms (0 = by size, -1 = one chain of launches; 1 / 2 = the pipelined prototypes of round 6, 3 = the tile-flow Cholesky).
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if any(k.startswith(("SR_FACT_", "SR_T64_")) for k in os.environ):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _lab  # noqa: F401,E402  (the switches exist in the lab build only)
from safe_exploration_amd import SimpleGPModel, workload, _lib  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1000, 2000, 5000, 10000]
    panels = [int(p) for p in os.environ.get("SR_PANELS", "0,1,2,4").split(",")]
    pipes = [int(p) for p in os.environ.get("SR_PIPE", "0").split(",")]
    n_s, n_u = int(os.environ.get("SR_NOUT", "2")), 1
    out = []
    for N in sizes:
        prob = workload.make_problem(4, N, n_s, n_u, 16)
        for P, pipe in [(P, q) for P in panels for q in pipes]:
            gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
            gp.set_fact_panel(P)
            gp.set_fact_pipeline(pipe)
            # Warm state and the MEDIAN of single refits.  (Round 3 timed five refits right after two warm-up calls on a model
            # created a moment before: at N = 5000 the table said 9.0 - 11.8 ms where `bench.py --workload c4 --n-train 5000`
            # -- three warm-up steps, twenty timed -- measures 5.0: the first refits of a new handle still touch fresh
            # scratch memory and run at a clock that has not ramped yet.)
            for _ in range(4 if N <= 12000 else 2):
                gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            torch.cuda.synchronize()
            reps = 9 if N <= 12000 else 2
            samples = []
            for _ in range(reps):
                t0 = time.perf_counter()
                gp.train(prob["Z"], prob["Y"], opt_hyp=False)
                torch.cuda.synchronize()
                samples.append(1e3 * (time.perf_counter() - t0))
            ms = sorted(samples)[len(samples) // 2]
            gp.prof_reset(); gp.prof_enable(True)
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            torch.cuda.synchronize()
            gp.prof_enable(False)
            split = {}
            for name, kid in (("gram", _lib.K_GRAM), ("potrf_diag", _lib.K_POTRF), ("chol_gemm", _lib.K_GEMM),
                              ("trinv_gemm", _lib.K_TRINV)):
                t, n = gp.prof_get(kid)
                split[name] = [round(t, 3), n]
            s2n = prob["noise_var"] + 1e-5 + 1e-8
            idx = np.random.default_rng(0).choice(N, min(N, 512), replace=False)
            mu, var = gp.predict(prob["Z"][idx])
            res = float(np.abs(mu + s2n[None, :] * gp.beta[idx] - prob["Y"][idx]).max())
            flops = n_s * (2.0 / 3.0) * float(N) ** 3
            rec = {"N": N, "n_out": n_s, "panel": P, "route": gp.fact_route(), "refit_ms": round(ms, 3), "TFLOPs": round(flops / ms / 1e9, 2),
                   "kernel_ms[total,launches]": split, "max|mu+s2n*alpha-y|": res}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del gp
    return out


if __name__ == "__main__":
    main()
