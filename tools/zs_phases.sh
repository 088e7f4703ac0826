#!/bin/bash
# Per-phase timing of the z-slide conv kernels on the GPU box (experiment build: CD_BUILD_TAG=exp
# CD_EXTRA_HIPCC_FLAGS=-DCD_ZS_EXPERIMENTS python -m calodiffusion_amd.build); differences between runs isolate the phases.
#   ping-pong kernel (zp_wave DBG bits): 1 no DMA/wait, 2 no conversion, 4 no reduce/store, 8 no MFMAs, 16 no fragment reads,
#   32 addresses prepared once, 64 no hand-over
out=gpurun_out/${1:-zsph}; mkdir -p $out
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_exp.so
for dbg in ${ZS_DBG_LIST:-0 1 3 4 7 39 8 16 24 64 88 127 0}; do
  echo "pp dbg=$dbg $(CD_ZS_DBG=$dbg timeout -k 5 90 python3 tools/conv_bench.py --iters 50 2>&1 | grep 'kernel us')" | tee -a $out/phases.log
done
