#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash scripts/evidence.sh r02
# Collects everything profiles/<tag>_* is made of into gpurun_out/ev_<tag>/ (bench lines, latency tables, kernel
# trace + PMC passes of the headline bench, kernel trace of the C4 model update).  Copy to profiles/ with
#   python scripts/summarize_rocpd.py gpurun_out/prof_<tag> profiles/<tag> c2p   and   cp gpurun_out/ev_<tag>/* profiles/
set -u
TAG=${1:-r05}
REPO=$(pwd)
EV=$REPO/gpurun_out/ev_$TAG
mkdir -p "$EV"
ulimit -c 0
line() { grep "^{" | tail -n 1; }
timeout 600 python bench.py 2>"$EV/.err" | line > "$EV/${TAG}_bench_c2p.json"
timeout 400 python bench.py --workload c2 2>>"$EV/.err" | line > "$EV/${TAG}_bench_c2.json"
timeout 400 python bench.py --workload c3 --steps 3 --warmup 1 2>>"$EV/.err" | line > "$EV/${TAG}_bench_c3.json"
timeout 400 python bench.py --workload c5 --steps 2 --warmup 1 2>>"$EV/.err" | line > "$EV/${TAG}_bench_c5.json"
timeout 600 python bench.py --workload c4 --steps 2 --warmup 1 2>>"$EV/.err" | line > "$EV/${TAG}_bench_c4.json"
timeout 300 python bench.py --workload c4 --n-train 5000 --steps 20 --warmup 3 2>>"$EV/.err" | line > "$EV/${TAG}_bench_refit5000.json"
timeout 600 python scripts/latency_grid.py > "$EV/${TAG}_latency_grid.txt" 2>>"$EV/.err"
timeout 300 python scripts/bal_ab.py 2000,3000,4000,5000 128,256,512,1024 > "$EV/${TAG}_bal_grid.txt" 2>>"$EV/.err"
timeout 300 python scripts/factor_bench.py > "$EV/${TAG}_factor_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/append_bench.py > "$EV/${TAG}_append_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/append_soak.py > "$EV/${TAG}_append_soak.txt" 2>>"$EV/.err"
timeout 600 python scripts/fuzz_append.py 30 150 > "$EV/${TAG}_fuzz_append.txt" 2>>"$EV/.err"
timeout 300 python scripts/chain_bench.py > "$EV/${TAG}_chain_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/onestep_bench.py > "$EV/${TAG}_onestep_bench.txt" 2>>"$EV/.err"
timeout 600 python scripts/fuzz_chain.py 60 4 > "$EV/${TAG}_fuzz_chain.txt" 2>>"$EV/.err"
timeout 600 python scripts/fuzz_predict.py > "$EV/${TAG}_fuzz_predict.txt" 2>>"$EV/.err"
timeout 600 python scripts/fuzz_server.py > "$EV/${TAG}_fuzz_server.txt" 2>>"$EV/.err"
timeout 300 python scripts/diag_bench.py > "$EV/${TAG}_diag_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/call_latency.py > "$EV/${TAG}_call_latency.txt" 2>>"$EV/.err"
# the journal experiments' kernels (defaultconfig_episode.py:39) on the same blocking single-query routes
for K in lin_mat52 mat52 lin_rbf; do
  timeout 300 python scripts/call_latency.py --kern $K >> "$EV/${TAG}_call_latency.txt" 2>>"$EV/.err"
done
timeout 200 python scripts/server_ticks.py > "$EV/${TAG}_server_ticks.txt" 2>>"$EV/.err"
timeout 300 python scripts/linearize_bench.py > "$EV/${TAG}_linearize_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/linearize_bench.py --kern lin_mat52 >> "$EV/${TAG}_linearize_bench.txt" 2>>"$EV/.err"
timeout 300 python scripts/exploration_step.py --kern lin_mat52 > "$EV/.expl_lin_mat52.txt" 2>>"$EV/.err"
timeout 400 python bench.py --workload c3s --steps 3 --warmup 1 2>>"$EV/.err" | line > "$EV/${TAG}_bench_c3s.json"
timeout 300 python bench.py --dry-nccl --steps 3 --warmup 1 --no-cpu-baseline 2>>"$EV/.err" | line > "$EV/${TAG}_bench_dry_nccl.json"
# the skeleton of the 128 x 128 main loop in isolation (built by scripts/build_bins.sh)
if [ -x scripts/_bin/pipe ]; then
  { echo "# hipcc --offload-arch=gfx950 -O3 scripts/mfma_pipe_tile.hip -o pipe && ./pipe   (1 x MI355X; modes in the header of the source)"; ./scripts/_bin/pipe; } > "$EV/${TAG}_mfma_pipe_tile.txt" 2>>"$EV/.err"
fi
# the 16-pivot chain of the diagonal-block kernel and the fp64 MFMA issue rate in isolation (built by scripts/build_bins.sh)
if [ -x scripts/_bin/pivot ]; then ./scripts/_bin/pivot > "$EV/${TAG}_pivot_chain.txt" 2>>"$EV/.err"; fi
if [ -x scripts/_bin/mfma_issue ]; then ./scripts/_bin/mfma_issue > "$EV/${TAG}_mfma_issue.txt" 2>>"$EV/.err"; fi
# the opt-in tile-flow Cholesky beside the chain of launches (same box), its lab sweep and its own diagnostics
{ bash scripts/flow_cmp.sh; SR_FLOW_STATS=1 timeout 200 python scripts/flow_check.py 2000 5000; } > "$EV/${TAG}_flow_cmp.txt" 2>>"$EV/.err"
timeout 300 python scripts/refit_ab.py 500,1000,2000,5000,10000 > "$EV/${TAG}_refit_sizes.txt" 2>>"$EV/.err"
timeout 300 python scripts/numpy_latency.py > "$EV/${TAG}_numpy_latency.txt" 2>>"$EV/.err"
timeout 300 python scripts/exploration_step.py > "$EV/${TAG}_exploration_step.txt" 2>>"$EV/.err"
cat "$EV/.expl_lin_mat52.txt" >> "$EV/${TAG}_exploration_step.txt" 2>/dev/null
timeout 300 python scripts/growing_model.py > "$EV/${TAG}_growing_model.txt" 2>>"$EV/.err"
timeout 900 bash scripts/profile_gpu.sh "$TAG" > "$EV/.profile.log" 2>&1
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$TAG/c4trace" -o c4 -- \
    python "$REPO/bench.py" --workload c4 --steps 1 --warmup 1 > "$EV/.c4trace.log" 2>&1 )
python - "$REPO/gpurun_out/prof_$TAG/c4trace" > "$EV/${TAG}_c4_kernel_stats.txt" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
print("# rocprofv3 --kernel-trace --stats  (python bench.py --workload c4 --steps 1 --warmup 1): two model updates at N=50000")
print("%-60s %8s %16s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in sqlite3.connect(dbs[0]).execute("select * from top_kernels"):
    print("%-60s %8d %16.0f %14.1f %8.3f" % (name[:60], calls, tot * 1.0, avg * 1.0, pct))
PY
# kernel timeline of ONE N = 5000 model update (who waits for whom) and kernel trace of the chain benchmark
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$REPO/gpurun_out/prof_$TAG/tl5000" -o tl -- \
    python "$REPO/scripts/timeline.py" run 5000 > "$EV/.tl.log" 2>&1 )
python scripts/timeline.py show "$REPO/gpurun_out/prof_$TAG/tl5000" 400 > "$EV/${TAG}_timeline5000.txt" 2>>"$EV/.err"
# kernel timeline of ONE N = 50000 model update: default build (long trailing updates on the unmasked stream) and with every
# trailing update on the masked stream; dispatches >= 4 ms, then the first chain-under-update region in full
{ echo "# kernel timeline of ONE model update at N = 50000, n_out = 2 (scripts/timeline.py run 50000 under rocprofv3 --kernel-trace; columns: start [ms], duration [ms], stream, grid threads, kernel; the short dispatches between two long ones collapsed per stream)"
  for ratio in default 0; do
    rm -rf "$REPO/gpurun_out/prof_$TAG/tl50000"
    if [ $ratio = default ]; then unset SR_FACT_FREE_RATIO; else export SR_FACT_FREE_RATIO=$ratio; fi
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$REPO/gpurun_out/prof_$TAG/tl50000" -o tl -- \
        python "$REPO/scripts/timeline.py" run 50000 > "$EV/.tl50k_$ratio.log" 2>&1 )
    python scripts/timeline.py showbig "$REPO/gpurun_out/prof_$TAG/tl50000" 1.0 > "$EV/.tl50k_$ratio.txt" 2>>"$EV/.err"
    echo "# ---- SR_FACT_FREE_RATIO=$ratio: dispatches >= 4 ms"
    grep -v "short dispatches" "$EV/.tl50k_$ratio.txt" | awk 'NR==1 || $2>4'
  done
  unset SR_FACT_FREE_RATIO
  echo "# ---- default build: from the end of the first panel's chain to the start of the second trailing update, everything"
  awk '$1>55 && $1<480' "$EV/.tl50k_default.txt" | head -150
  rm -rf "$REPO/gpurun_out/prof_$TAG/tl50000"; } > "$EV/${TAG}_timeline50000.txt"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$TAG/chain" -o chain -- \
    python "$REPO/scripts/chain_bench.py" > "$EV/.chaintrace.log" 2>&1 )
python - "$REPO/gpurun_out/prof_$TAG/chain" > "$EV/${TAG}_chain_kernel_stats.txt" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
print("# rocprofv3 --kernel-trace --stats  (python scripts/chain_bench.py): persistent chain kernel against per-step launches")
print("%-70s %8s %14s %12s" % ("kernel", "calls", "total_us", "avg_us"))
for name, calls, tot, avg, pct in sqlite3.connect(dbs[0]).execute("select * from top_kernels"):
    if "chain" in name or "small" in name or "ellipsoid" in name:
        print("%-70s %8d %14.0f %12.1f" % (name[:70], calls, tot * 1.0, avg * 1.0))
PY
# per-kernel GPU time of the small-batch predict path at N = 5000
for T in 1 16 32 64 128; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$TAG/lat/t$T" -o lat -- \
      python "$REPO/scripts/latency_trace.py" run 5000 $T > "$EV/.lat_$T.log" 2>&1 )
done
{ echo "# per-kernel GPU time of predict at N = 5000 (rocprofv3 --kernel-trace --stats, 210 calls each; T = directory name)";
  python scripts/latency_trace.py show "$REPO/gpurun_out/prof_$TAG/lat"; grep -h wall "$EV"/.lat_*.log; } > "$EV/${TAG}_latency_kernels.txt" 2>>"$EV/.err"
ls -la "$EV"
tail -n 5 "$EV/.err"
