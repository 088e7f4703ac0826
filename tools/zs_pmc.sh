#!/bin/bash
# Instruction counts / busy cycles of the z-slide conv kernels (rocprofv3 PMC, one pass per counter group; conv_bench launches)
#   bash tools/zs_pmc.sh <tag>      (CD_ZS_V1=1 in the environment profiles the matrix/helper-wave form)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-pmc}; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace -d $out/g$i -o pmc -- python3 tools/conv_bench.py --iters 5 ${ZS_BENCH_ARGS} >> $out/log.txt 2>&1
  echo "group $i rc=$?" | tee -a $out/log.txt
done
python3 tools/pmc_read.py $out zslide | tee $out/summary.txt
