#!/usr/bin/env python3
"""Per-kernel breakdown of the small-batch path.  Run under the profiler ON THE GPU BOX:

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/lat -o lat -- \
        python $REPO/scripts/latency_trace.py run 5000 1
    python scripts/latency_trace.py show gpurun_out/lat

`run N T [H]` issues 200 predict calls of T queries against an N-point pendulum model (plus wall time per
call; with H also 50 H-step reachability chains over T rollouts); `show DIR` prints the top_kernels view of the rocpd database rocprofv3 left in DIR."""
import glob
import os
import sqlite3
import sys
import time


def run(N, T, H=1):
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from safe_exploration_amd import SimpleGPModel, workload, _buffers as B
    prob = workload.make_problem(9, N, 2, 1, max(T, 8), sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"], prob["Y"], opt_hyp=False)
    if os.environ.get("SR_SMALL_PATH"):          # A/B of the routes: 1 default, 2 without the one-launch pass, 0 plain tiles
        gp.set_small_path(int(os.environ["SR_SMALL_PATH"]))
    x = B.as_dev(np.hstack((prob["p"][:T], prob["k_ff"][:T])), gp.device)
    for _ in range(10):
        gp.predict_device(x, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        gp.predict_device(x, True)
    torch.cuda.synchronize()
    print("N=%d T=%d wall per predict call: %.1f us" % (N, T, (time.perf_counter() - t0) / 200 * 1e6))
    if H > 1:
        from safe_exploration_amd import gp_reachability as reach
        roll = workload.random_rollout_controls(3, T, H, 2, 1)
        tr = {k: B.as_dev(v, gp.device) for k, v in roll.items()}
        l, a, b = np.array([0.05, 0.02]), 0.8 * np.eye(2), np.zeros((2, 1))
        f = lambda: reach.multistep_reachability_batch(tr["p0"], gp, tr["k_fb"], tr["k_ff"], l, l, None, 2.0, a, b)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            f()
        torch.cuda.synchronize()
        print("N=%d T=%d H=%d wall per multistep call: %.1f us" % (N, T, H, (time.perf_counter() - t0) / 50 * 1e6))


def show(d):
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        print("==", db)
        con = sqlite3.connect(db)
        print("%-64s %7s %12s %10s" % ("kernel", "calls", "total_us", "avg_us"))
        for name, calls, tot, avg, pct in con.execute("select * from top_kernels"):
            if calls >= 50:
                print("%-64s %7d %12.0f %10.2f" % (name[:64], calls, tot, avg))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    else:
        show(sys.argv[2])
