#!/usr/bin/env python3
"""Turn the rocpd SQLite databases rocprofv3 (ROCm 7.2) writes into the text summaries committed
under profiles/:   python scripts/summarize_rocpd.py gpurun_out/prof_r01 profiles/r01 c2p

  <prefix>_kernel_stats.txt   per-kernel calls / total / average duration (== `--kernel-trace --stats`)
  <prefix>_pmc.txt            per-kernel per-counter totals and per-launch means, one block per PMC pass
  profiles/pmc_<workload>.json  HBM-side bytes per sr_var_kernel launch, read by bench.py (`roofline.traffic`)

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128 B request for wide (16 B/lane) coalesced reads -> doubled.  Infinity-Cache
hits are included in the counter (fabric-side, not DRAM-side, bytes).
"""
import glob
import json
import os
import sqlite3
import sys


def main():
    src, prefix, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    tdb = glob.glob(os.path.join(src, "trace", "*.db"))
    with open(prefix + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats  (python bench.py --steps 5 --warmup 1 --no-cpu-baseline)\n")
        f.write("# source: %s (view top_kernels); durations in microseconds\n" % tdb[0])
        f.write("%-60s %8s %16s %14s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        db = sqlite3.connect(tdb[0])
        for name, calls, tot, avg, pct in db.execute("select * from top_kernels"):
            f.write("%-60s %8d %16.0f %14.1f %8.3f\n" % (name[:60], calls, tot * 1.0, avg * 1.0, pct))
    per = {}
    with open(prefix + "_pmc.txt", "w") as f:
        f.write("# rocprofv3 --pmc <counters> --kernel-trace, one pass per line group "
                "(python bench.py --steps 2 --warmup 1 --no-cpu-baseline)\n")
        for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
            dbs = glob.glob(os.path.join(d, "*.db"))
            if not dbs:
                continue
            f.write("\n## pass %s\n" % os.path.basename(d))
            f.write("%-48s %-28s %8s %20s %20s\n" % ("kernel", "counter", "launches", "total", "per_launch"))
            db = sqlite3.connect(dbs[0])
            q = ("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                 "group by kernel_name, counter_name order by sum(value) desc")
            for kn, cn, n, tot, avg in db.execute(q):
                if not kn.startswith(("sr_", "void sr_")):
                    continue
                f.write("%-48s %-28s %8d %20.6g %20.6g\n" % (kn[:48], cn, n, tot, avg))
                per[(kn.split("(")[0].replace("void ", "").split("<")[0], cn)] = avg
    fetch = per.get(("sr_var_kernel", "FETCH_SIZE"))
    write = per.get(("sr_var_kernel", "WRITE_SIZE"))
    if fetch is not None:
        hbm = (2.0 * fetch + (write or 0.0)) * 1024.0
        out = {"workload": workload, "sr_var_kernel_hbm_bytes_per_launch": hbm,
               "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
               "note": "fabric-side bytes (Infinity-Cache hits included); FETCH_SIZE doubled per the gfx950 "
                       "wide-read correction of MI355X_MICROARCH.md",
               "source": os.path.basename(prefix) + "_pmc.txt"}
        with open(os.path.join(os.path.dirname(prefix) or ".", "pmc_%s.json" % workload), "w") as f:
            json.dump(out, f, indent=1)
        print(out)


if __name__ == "__main__":
    main()
