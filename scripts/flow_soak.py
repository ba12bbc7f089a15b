"""Soak of the opt-in tile-flow Cholesky: model updates of random sizes, output counts and panel widths in ONE process, each
compared with the chain of launches on the same data (alpha, U^-1 to rounding; the posterior identity), interleaved with
predicts and one-point appends on the same handle, two handles alive at a time.  Reports how every update ran (route 4 = tile
flow) -- a fall back to launches (a wait that ran into its time-out) counts as a failure here.
usage (GPU box): python scripts/flow_soak.py [updates] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402


def main():
    n_upd = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, routes, worst = 0, {}, 0.0
    t0 = time.perf_counter()
    keep = None
    for it in range(n_upd):
        N = int(rng.choice([300, 390, 700, 1100, 1500, 1930, 2600, 3300, 4100, 5000, 6500, 9000]))
        if it % 20 == 19:
            N = 12500
        n_s = int(rng.choice([1, 2, 2, 3, 4])) if N <= 6500 else 2
        panel = int(rng.choice([0, 0, 2, 3, 4, 6, 8]))
        prob = workload.make_problem(1000 + it, N, n_s, 1, 8)
        ref = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        ref.set_fact_pipeline(-1)
        ref.train(prob["Z"], prob["Y"], opt_hyp=False)
        a0, w0 = ref.export_state()
        gp = SimpleGPModel(n_s, n_s, 1, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
        gp.set_fact_panel(panel)
        gp.set_fact_pipeline(3)
        for rep in range(int(rng.integers(1, 4))):
            gp.train(prob["Z"], prob["Y"], opt_hyp=False)
            r = gp.fact_route()
            routes[r] = routes.get(r, 0) + 1
            a1, w1 = gp.export_state()
            da = float((a1 - a0).abs().max() / a0.abs().max())
            dw = float((w1 - w0).abs().max() / w0.abs().max())
            worst = max(worst, da, dw)
            ok = r == 4 and da < 1e-9 and dw < 1e-10
            if rep == 0 and N <= 5000:
                # a predict and a one-point append between two updates
                mu, _ = gp.predict(prob["Z"][:64])
                s2n = prob["noise_var"] + 1e-5 + 1e-8
                ok = ok and float(np.abs(mu + s2n[None, :] * gp.beta[:64] - prob["Y"][:64]).max()) < 1e-8
                gp.update_model(prob["Z"][:1] + 0.01, prob["Y"][:1], opt_hyp=False)
            if not ok:
                bad += 1
                print(f"FAIL it={it} N={N} n_out={n_s} panel={panel} rep={rep} route={r} d_alpha={da:.2e} d_Uinv={dw:.2e}", flush=True)
        del ref, a0, w0, a1, w1
        keep = gp                     # the previous flow handle stays alive through the next iteration
        if it % 10 == 9:
            print(f"  {it + 1} models, routes {routes}, worst rel. difference {worst:.2e}, {time.perf_counter() - t0:.0f} s", flush=True)
            torch.cuda.empty_cache()
    del keep
    print(f"{n_upd} models, updates by route {routes}, failures {bad}, worst rel. difference {worst:.2e}, {time.perf_counter() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
