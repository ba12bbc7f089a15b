#!/usr/bin/env python3
"""The inner loop of the reference's exploration runner, per step (exploration_runner.py:176-189): one query through
predict (NumPy in, NumPy out), one observed transition appended with update_model(replace_old=False), the information
gain.  Wall time per call and per step, by model size.  GPU box:  python scripts/exploration_step.py [Ns] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_exploration_amd import SimpleGPModel, workload  # noqa: E402


def main():
    args = list(sys.argv[1:])
    kt = "rbf"
    if "--kern" in args:                                    # rbf (default) | mat52 | lin_rbf | lin_mat52
        i = args.index("--kern")
        kt = args[i + 1]
        del args[i:i + 2]
    Ns = [int(v) for v in (args[0] if len(args) > 0 else "50,200,1000,5000").split(",")]
    steps = int(args[1]) if len(args) > 1 else 60
    print("# kern_types = ['%s'] * 2" % kt)
    print("%6s %12s %14s %14s %12s %12s %10s   [us per call: MEDIAN of %d steps after 5 warm-up steps; mean step and worst step beside it --"
          % ("N", "predict(1)", "update(+1)", "info gain", "step", "mean step", "max step", steps))
    print("#  the mean carries the steps at which the padded size of the model grows (one per 128 points: reallocation, and the first launch"
          "\n#  of a kernel instantiation the process has not used yet, ~1.5 ms once per process)]")
    for N in Ns:
        prob = workload.make_problem(6, N + steps + 5, 2, 1, 8)
        if kt == "rbf":
            hyp = workload.hyp_list(prob)
        else:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from call_latency import kern_hyp
            hyp = kern_hyp(kt, np.random.default_rng(N), 3, 2)
        gp = SimpleGPModel(2, 2, 1, kern_types=[kt] * 2, hyp=hyp, device="cuda:0")
        Z, Y = prob["Z"], prob["Y"]
        gp.train(Z[:N], Y[:N], opt_hyp=False)
        rec = []
        for i in range(steps + 5):
            z_i, y_i = Z[N + i:N + i + 1], Y[N + i:N + i + 1]
            t0 = time.perf_counter()
            mu, s2 = gp.predict(z_i)
            t1 = time.perf_counter()
            gp.update_model(z_i, y_i, opt_hyp=False, replace_old=False)
            t2 = time.perf_counter()
            ig = gp.information_gain()
            t3 = time.perf_counter()
            if i >= 5:
                rec.append((t1 - t0, t2 - t1, t3 - t2))
        assert np.all(np.isfinite(mu)) and np.all(s2 > 0) and np.all(np.isfinite(ig))
        rec = np.array(rec) * 1e6
        med, tot = np.median(rec, axis=0), rec.sum(axis=1)
        print("%6d %12.1f %14.1f %14.1f %12.1f %12.1f %10.1f" % (N, med[0], med[1], med[2], np.median(tot), tot.mean(), tot.max()), flush=True)


if __name__ == "__main__":
    main()
