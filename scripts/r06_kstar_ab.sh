#!/bin/bash
OUT=gpurun_out/r06; mkdir -p $OUT
for adj in 0 1 2 0 1 2; do
  SR_KSTAR_ADJ=$adj timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['roofline_kstar']
print('kstar form $adj  value %.4e evals/s  ms/step %.3f  sr_var %.3f ms %.2f TF  kstar %.3f ms %.0f GB/s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], k['avg_launch_ms'], k['achieved']))"
done | tee $OUT/kstar_ab.txt
SR_KSTAR_ADJ=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "predict or onestep or headline" 2>&1 | tail -2
