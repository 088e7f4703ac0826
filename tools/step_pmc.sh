#!/bin/bash
# PMC counters of EVERY kernel of eager denoise steps, three passes (instruction mix, busy / wait cycles, LDS): which kernels are bound
# by what.   bash tools/step_pmc.sh <tag> <kernel substring> [<kernel substring> ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $grp --kernel-trace -d $out/g$i -o pmc -- python3 tools/step_breakdown.py --reps 3 ${STEP_ARGS} >> $out/log.txt 2>&1
  echo "group $i rc=$?" | tee -a $out/log.txt
done
for pat in "$@"; do echo "=== $pat"; python3 tools/pmc_read.py $out "$pat"; done | tee $out/summary.txt
find $out -name "*.db" -delete
