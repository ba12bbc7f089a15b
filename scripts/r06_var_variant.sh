#!/bin/bash
# round 6: variant 5 of sr_var_kernel (pairs of row blocks per workgroup) against variant 4, groups swept; GPU box, repo root
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants_agree" 2>&1 | tail -2
for v in 4 5; do for g in ${GROUPS_:-32 64 128}; do
  timeout 200 python bench.py --var-variant $v --var-group $g --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant $v var_group %4d  value %.4e evals/s  ms/step %.3f  sr_var %.3f ms %.2f TF' % ($g, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved']))"
done; done | tee $OUT/var_variant5.txt
