#!/bin/bash
# Instruction counts / busy cycles / LDS conflicts of the 3x3x3 weight-gradient kernels (rocprofv3 PMC, one pass per counter group; level-0
# shape through tools/wgrad_bench.py):  bash tools/wgrad_pmc.sh <tag>   -- the z-sliding kernel, then (CD_NO_WGRAD_RING=1) round 4's two-plane units
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-wgpmc}; rm -rf $out; mkdir -p $out
for arm in ring old; do
  if [ $arm = old ]; then export CD_NO_WGRAD_RING=1; pat=wgrad_f16x2_kernel; else unset CD_NO_WGRAD_RING; pat=wgrad_ring; fi
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace -d $out/${arm}_g$i -o pmc -- python3 tools/wgrad_bench.py --iters 5 >> $out/log.txt 2>&1
    echo "$arm group $i rc=$?" | tee -a $out/log.txt
  done
  echo "== $arm ($pat)" | tee -a $out/summary.txt
  for i in 1 2 3; do python3 tools/pmc_read.py $out/${arm}_g$i $pat | tee -a $out/summary.txt; done
  rm -rf $out/${arm}_g*
done
