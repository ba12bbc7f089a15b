#!/usr/bin/env python3
"""Kernel timeline of row appends.  ON THE GPU BOX:
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $REPO/gpurun_out/apptl -o app -- python $REPO/scripts/append_trace.py run 5000 1
    python scripts/append_trace.py show gpurun_out/apptl"""
import glob
import os
import sqlite3
import sys


def run(N, m):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from safe_exploration_amd import SimpleGPModel, workload
    prob = workload.make_problem(13, N + 4 * m, 2, 1, 8, sf2=0.01)
    gp = SimpleGPModel(2, 2, 1, kern_types=["rbf"] * 2, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.train(prob["Z"][:N], prob["Y"][:N], opt_hyp=False)
    torch.cuda.synchronize()
    for r in range(4):
        lo = N + r * m
        gp.update_model(prob["Z"][lo:lo + m], prob["Y"][lo:lo + m], opt_hyp=False, replace_old=False)
        torch.cuda.synchronize()


def show(d):
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        rows = list(con.execute("select name, start, end from kernels order by start"))
        # the last append starts at the last sr_append_y_kernel
        st = [i for i, r in enumerate(rows) if "sr_append_y_kernel" in r[0]]
        rows = rows[max(st[-1] - 3, 0):] if st else rows
        t0 = rows[0][1]
        print("last append: %d dispatches over %.1f us, kernel time %.1f us" % (
            len(rows), (rows[-1][2] - t0) / 1e3, sum(r[2] - r[1] for r in rows) / 1e3))
        for name, a, b in rows:
            print("%9.1f %8.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, name.split("(")[0][-60:]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        show(sys.argv[2])
