"""Tile-flow Cholesky (sr_flow.hip) against the chain of launches: same model, both routes -- alpha and U^-1 compared, the
residual of K alpha = y, and the median refit time of either.  usage (GPU box): python scripts/flow_check.py [N ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if any(k.startswith("SR_FLOW_") and k not in ("SR_FLOW_STATS", "SR_FLOW_ONLY") for k in os.environ):
    import _lab  # noqa: F401,E402  (SR_FLOW_PANEL / _BAND / _LOOK / _ACQ exist in the lab build only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402


def fit(prob, n_s, n_u, pipe, reps):
    gp = SimpleGPModel(n_s, n_s, n_u, kern_types=["rbf"] * n_s, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.set_fact_pipeline(pipe)
    for _ in range(3):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    torch.cuda.synchronize()
    samples = []
    for _ in range(reps):
        t0 = time.perf_counter()
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        torch.cuda.synchronize()
        samples.append(1e3 * (time.perf_counter() - t0))
    _, wt = gp.export_state()
    return gp, gp.beta.copy(), wt.clone(), sorted(samples)[len(samples) // 2]


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [700, 2000, 5000]
    n_s, n_u = int(os.environ.get("SR_NOUT", "2")), 1
    for N in sizes:
        prob = workload.make_problem(4, N, n_s, n_u, 16)
        if os.environ.get("SR_FLOW_ONLY"):
            gp1, b1, w1, ms1 = fit(prob, n_s, n_u, 3, 7)
            print(f"N={N} route {gp1.fact_route()} flow {ms1:.3f} ms  " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SR_FLOW_")), flush=True)
            del gp1, w1
            continue
        gp0, b0, w0, ms0 = fit(prob, n_s, n_u, -1, 7)
        r0 = gp0.fact_route()
        del gp0
        gp1, b1, w1, ms1 = fit(prob, n_s, n_u, 3, 7)
        r1 = gp1.fact_route()
        s2n = prob["noise_var"] + 1e-5 + 1e-8
        idx = np.random.default_rng(0).choice(N, min(N, 512), replace=False)
        mu, _ = gp1.predict(prob["Z"][idx])
        res = float(np.abs(mu + s2n[None, :] * gp1.beta[idx] - prob["Y"][idx]).max())
        db = float(np.abs(b1 - b0).max() / np.abs(b0).max())
        dw = float((w1 - w0).abs().max() / w0.abs().max())
        if os.environ.get("SR_FLOW_STATS"):
            import ctypes
            from safe_exploration_amd import _lib
            nb = gp1._handle.Np // 128
            buf = (ctypes.c_uint * (24 + n_s * nb))()
            n = _lib.lib.sr_gp_flow_stats(gp1._handle.h, buf, len(buf))
            if n > 0:
                names = ["upd next panel", "upd behind", "diag tiles", "near upd", "near solve", "far block"]
                for k, nm in enumerate(names):
                    c, t, w = buf[4 * k], buf[4 * k + 1], buf[4 * k + 2]
                    if c:
                        print(f"   {nm:15s} {c:6d} tasks  {t / c / 100:8.1f} us each  of which waiting {w / c / 100:8.1f}", flush=True)
                tk = [buf[24 + k] / 100.0 for k in range(nb)]
                print("   diagonal block done at us:", " ".join(f"{x:.0f}" for x in tk), flush=True)
                print("   steps us:", " ".join(f"{b - a:.0f}" for a, b in zip([0.0] + tk[:-1], tk)), flush=True)
        print(f"N={N} routes {r0}/{r1}  launches {ms0:.3f} ms  flow {ms1:.3f} ms  rel|d alpha| {db:.2e}  rel|d U^-1| {dw:.2e}  "
              f"residual {res:.2e}", flush=True)
        del gp1, w0, w1
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
