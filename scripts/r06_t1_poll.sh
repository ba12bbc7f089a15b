#!/bin/bash
# round 6: polling finaliser of the streamed single-query kernel against its two ticket levels (lab build: SR_ST1_POLL)
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream or small_batch or call1 or single or latency or predict" 2>&1 | tail -3
for v in 1 0 1 0; do SR_ST1_POLL=$v timeout 300 python scripts/t1_probe.py 600,1000,2000,3000,5000,8000 | sed "s/^/poll=$v /"; done | tee $OUT/t1_poll.txt
timeout 300 python scripts/call_latency.py 2>&1 | grep -E "N= *(1000|2000|5000) " | tee -a $OUT/t1_poll.txt
