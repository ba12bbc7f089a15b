#!/usr/bin/env python3
"""Kernel timeline of ONE model update (who waits for whom).  ON THE GPU BOX:

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d $REPO/gpurun_out/tl -o tl -- python $REPO/scripts/timeline.py run 5000
    python scripts/timeline.py show gpurun_out/tl [max_rows]

`show` prints, for the LAST update of the run, every dispatch with start offset / duration / stream and a summary of the
busy time per stream and of the gaps on the critical stream."""
import glob
import os
import sqlite3
import sys


def run(N, n_out=2, panel=0):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from safe_exploration_amd import SimpleGPModel, workload
    prob = workload.make_problem(4, N, n_out, 1, 16)
    gp = SimpleGPModel(n_out, n_out, 1, kern_types=["rbf"] * n_out, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.set_fact_panel(panel)
    for _ in range(3):
        gp.train(prob["Z"], prob["Y"], opt_hyp=False)
        torch.cuda.synchronize()


def show(d, max_rows=400):
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        if not cols:
            print("views:", [r[0] for r in con.execute("select name from sqlite_master")])
            continue
        want = [c for c in ("name", "start", "end", "stream_id", "queue_id", "stream", "queue", "grid_x", "grid_size", "grid_size_x")
                if c in cols]
        print("== %s  columns: %s" % (db, cols))
        rows = list(con.execute("select %s from kernels order by start" % ", ".join(want)))
        ix = {c: i for i, c in enumerate(want)}
        # the last update starts at the last sr_gram / pack kernel group: find the last 'sr_pack_y' launch
        starts = [i for i, r in enumerate(rows) if "sr_pack_y" in r[ix["name"]]]
        rows = rows[starts[-1]:] if starts else rows
        t0 = rows[0][ix["start"]]
        skey = "stream_id" if "stream_id" in ix else ("stream" if "stream" in ix else ("queue_id" if "queue_id" in ix else None))
        busy = {}
        for r in rows:
            s = r[ix[skey]] if skey else 0
            busy[s] = busy.get(s, 0) + (r[ix["end"]] - r[ix["start"]])
        print("update spans %.1f us, %d dispatches; busy us per stream: %s" %
              ((max(r[ix["end"]] for r in rows) - t0) / 1e3, len(rows), {k: round(v / 1e3, 1) for k, v in busy.items()}))
        for r in rows[:max_rows]:
            g = r[ix["grid_x"]] if "grid_x" in ix else (r[ix["grid_size_x"]] if "grid_size_x" in ix else (r[ix["grid_size"]] if "grid_size" in ix else 0))
            print("%9.1f %8.1f  s=%-4s g=%-7s %s" % ((r[ix["start"]] - t0) / 1e3, (r[ix["end"]] - r[ix["start"]]) / 1e3,
                                                  r[ix[skey]] if skey else "-", g, r[ix["name"]].split("(")[0][-44:]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 400)
