#!/bin/bash
# round 6: sweep of the scheduling group of sr_var_kernel (query tiles per group) with the pipelined main loop; GPU box, repo root
OUT=gpurun_out/r06; mkdir -p $OUT
for g in 8 16 32 48 64 96 128 256; do
  timeout 200 python bench.py --var-group $g --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('var_group %4d  value %.4e evals/s  ms/step %.3f  sr_var %.3f ms %.2f TF  kstar %.3f ms' % ($g, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], d['roofline_kstar']['avg_launch_ms']))"
done | tee $OUT/var_group_sweep.txt
