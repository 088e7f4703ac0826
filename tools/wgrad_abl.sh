#!/bin/bash
# bash tools/wgrad_abl.sh: phases of wgrad_ring_f16x2_kernel switched off one at a time (experiment builds libcalodiff_hip_wabl.so:
# -DCD_WGRAD_ABL=0, run-time switches CD_WGRAD_ABL = 2 no K loop | 4 no prefetch of the next unit | 8 no partial write; _wabl1.so: no MFMAs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/wabl
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_wabl.so
for abl in 0 2 4 8 6 14 0; do
  CD_WGRAD_ABL=$abl timeout -k 10 120 python tools/wgrad_bench.py "$@" 2>/dev/null | tee -a gpurun_out/wabl/abl.log
done
CD_NO_WGRAD_RING=1 timeout -k 10 120 python tools/wgrad_bench.py "$@" 2>/dev/null | tee -a gpurun_out/wabl/abl.log
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_wabl1.so
echo "no MFMAs:" | tee -a gpurun_out/wabl/abl.log
timeout -k 10 120 python tools/wgrad_bench.py "$@" 2>/dev/null | tee -a gpurun_out/wabl/abl.log
