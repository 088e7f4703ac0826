#!/bin/bash
# In-kernel cycle stamps of the one-wave-per-SIMD z-slide kernel (experiment build, CD_ZS_DBG=2048): per-wave mean cycles of the
# parts of a step (load wait, convert, reduce + store, tap addresses, MFMAs, barrier), plain and normalised input.
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_exp.so
out=gpurun_out/${1:-stamps}; mkdir -p $out
for norm in ""; do
  echo "== stamps $norm"
  CD_ZS_DBG=2048 timeout -k 5 90 python3 tools/conv_bench.py --iters 12 $norm 2>&1 | grep -E "z3 stamps|kernel us"
done 2>&1 | tee $out/stamps.log
unset CALODIFF_LIB
