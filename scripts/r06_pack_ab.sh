#!/bin/bash
# round 6: several work items per workgroup in the streamed MFMA kernel at 64 columns (default: more items than CUs are dealt to one
# workgroup per CU, longest first) against one item per workgroup (SR_ST_NO_PACK=1, the plan of the first half of the round); lab build.
# bash scripts/r06_pack_ab.sh > gpurun_out/pack_ab.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or small_batch_routes or streamed or stream" 2>&1 | tail -2
for pass in 1 2; do
  for v in 1 0; do
    echo "SR_ST_NO_PACK=$v"
    SR_ST_NO_PACK=$v timeout 600 python scripts/latency_scan.py 2600 10200 400 48,64
  done
done
