set -u
bash scripts/profile_gpu.sh r01d > gpurun_out/prof_r01d.log 2>&1
python bench.py > gpurun_out/bench_r01d_c2p.json 2> gpurun_out/bench_err.log; tail -1 gpurun_out/bench_r01d_c2p.json | cut -c1-400
for w in c2 c3; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r01d_$w.json; cut -c1-300 gpurun_out/bench_r01d_$w.json; done
python __graft_entry__.py smoke 2>&1 | tail -2
