#!/bin/bash
# A/B of the z-slide convolution forms on the GPU box: parity tests of the conv kernels, then the level-0 conv micro-benchmark
# with the one-wave-per-SIMD kernel (default) and the matrix/helper-wave kernel (CD_ZS_V1=1), Dataset-2 / Dataset-3 / HGCal shapes.
out=gpurun_out/${1:-zsab}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv or denoise or strip or resnet" > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for v in sw v1; do
  if [ $v = v1 ]; then export CD_ZS_V1=1; else unset CD_ZS_V1; fi
  echo "== $v" | tee -a $out/bench.log
  timeout -k 10 120 python tools/conv_bench.py --iters 30 2>&1 | tail -2 | tee -a $out/bench.log || exit 1
  timeout -k 10 120 python tools/conv_bench.py --iters 20 --batch 32 --dims 45,50,18 2>&1 | tail -2 | tee -a $out/bench.log || exit 1
  timeout -k 10 120 python tools/conv_bench.py --iters 20 --batch 16 --dims 28,12,21 2>&1 | tail -2 | tee -a $out/bench.log || exit 1
done
exit $rc
