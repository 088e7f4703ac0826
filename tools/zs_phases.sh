#!/bin/bash
# per-phase timing of the z-slide conv kernel (CD_ZS_DBG switches); differences between runs isolate the phases
mkdir -p gpurun_out
for dbg in ${ZS_DBG_LIST:-0 2 4 6 16 0}; do
  echo "dbg=$dbg $(CD_ZS_DBG=$dbg timeout -k 5 90 python3 tools/conv_bench.py --iters 50 2>&1 | grep 'kernel us')" | tee -a gpurun_out/zs_phases.log
done
