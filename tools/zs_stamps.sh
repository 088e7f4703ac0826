#!/bin/bash
# In-kernel cycle stamps of the ping-pong z-slide kernel (experiment build, DBG & 2048): per-wave mean cycles of the matrix
# phases, the support phases and the barrier waits; then the A/B timing of the product build.
export CALODIFF_LIB=$PWD/calodiffusion_amd/lib/libcalodiff_hip_exp.so
out=gpurun_out/${1:-stamps}; mkdir -p $out
for dbg in ${ZS_DBG_LIST:-2048 2055 2136}; do
  echo "== dbg $dbg"
  CD_ZS_DBG=$dbg timeout -k 5 90 python3 tools/conv_bench.py --iters 12 2>&1 | grep -E "zp stamps|zp fine|kernel us" | grep -E "wg   0 wave [04]|kernel|fine"
done 2>&1 | tee $out/stamps.log
unset CALODIFF_LIB
[ -n "$ZS_NO_AB" ] || bash tools/zs_ab.sh ${1:-stamps}
