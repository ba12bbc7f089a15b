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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if any(k.startswith(("SR_FACT_", "SR_T64_")) for k in os.environ):
        import _lab  # noqa: F401  (the switches exist in the lab build only)
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from safe_exploration_amd import SimpleGPModel, workload
    prob = workload.make_problem(4, N, n_out, 1, 16)
    gp = SimpleGPModel(n_out, n_out, 1, kern_types=["rbf"] * n_out, hyp=workload.hyp_list(prob), device="cuda:0")
    gp.set_fact_panel(panel)
    if os.environ.get("SR_PIPE"):
        gp.set_fact_pipeline(int(os.environ["SR_PIPE"]))        # 0: one chain, 1 / 2: the pipelined forms of round 6
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


def showbig(d, min_ms=1.0):
    """For an update of a LARGE model: every dispatch of at least min_ms on its own line, the short ones between them
    collapsed per stream into (first start, last end, count, busy time)."""
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        if not cols:
            continue
        skey = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "name")
        gkey = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "start")
        rows = list(con.execute("select name, start, end, %s, %s from kernels order by start" % (skey, gkey)))
        starts = [i for i, r in enumerate(rows) if "sr_pack_y" in r[0]]
        rows = rows[starts[-1]:] if starts else rows
        t0 = rows[0][1]
        print("== %s: last update spans %.2f ms, %d dispatches" % (db, (max(r[2] for r in rows) - t0) / 1e6, len(rows)))
        pend = {}

        def flush(sid):
            g = pend.pop(sid, None)
            if g:
                print("%10.2f %9.2f  s=%-4s   [%d short dispatches, busy %.2f ms]" % ((g[0] - t0) / 1e6, (g[1] - g[0]) / 1e6, sid, g[2], g[3] / 1e6))
        for name, st, en, sid, grid in rows:
            if en - st >= min_ms * 1e6:
                flush(sid)
                print("%10.2f %9.2f  s=%-4s g=%-8s %s" % ((st - t0) / 1e6, (en - st) / 1e6, sid, grid, name.split("(")[0][-44:]))
            else:
                g = pend.get(sid)
                pend[sid] = [st, en, 1, en - st] if g is None else [g[0], max(g[1], en), g[2] + 1, g[3] + en - st]
        for sid in list(pend):
            flush(sid)


if __name__ == "__main__":
    if sys.argv[1] == "showbig":
        showbig(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 400)
