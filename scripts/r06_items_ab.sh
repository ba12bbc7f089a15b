#!/bin/bash
# round 6: work items of the streamed MFMA kernel planned for the shortest launch (default) against round 5's runs of whole chunks (SR_ST_KR=-1)
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or small_batch_routes or streamed" 2>&1 | tail -2
for v in 0 -1 0 -1; do SR_ST_KR=$v timeout 300 python scripts/bal_ab.py 2500,3500,4000,5000,6000,8000,10000 16,32,64 | sed "s/^SR_BAL_WGS=- SR_BAL_THR=- /kr=$v /"; done | tee $OUT/items_ab.txt
