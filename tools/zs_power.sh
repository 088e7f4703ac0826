#!/bin/bash
# Is the z-slide conv clock/power bound?  The SAME workgroup program (4 chunks per sample, 26 steps each) on 64 / 128 / 192 / 256
# CUs (batch 16 / 32 / 48 / 64, one workgroup per CU; experiment build, CD_ZS_NCHUNK): if the per-launch time grows with the
# number of busy CUs, the chip is lowering its clock under the load and no rearrangement of the instruction stream will help.
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_exp.so
out=gpurun_out/${1:-power}; mkdir -p $out
for v in sw v1; do
  if [ $v = v1 ]; then export CD_ZS_V1=1; else unset CD_ZS_V1; fi
  for b in 8 16 32 48 64; do
    echo "== $v batch $b" | tee -a $out/power.log
    CD_ZS_NCHUNK=4 timeout -k 10 120 python tools/conv_bench.py --iters 40 --batch $b 2>&1 | tail -2 | tee -a $out/power.log || exit 1
  done
done
