#!/bin/bash
# attention passes at Dataset-2 level 0 (batch 64) for one and two grids' worth of voxels: fixed vs per-tile cost
out=gpurun_out/${1:-attn}; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "attention or denoise" > $out/tests.log 2>&1; rc=$?
tail -2 $out/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for dims in 45,16,9 90,16,9 23,8,4; do
  echo "== dims $dims" | tee -a $out/attn.log
  timeout -k 10 120 python tools/attn_bench.py --dims $dims 2>&1 | tail -6 | tee -a $out/attn.log || exit 1
done
