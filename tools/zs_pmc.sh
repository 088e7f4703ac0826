#!/bin/bash
# PMC counters of the z-slide conv kernel (one rocprofv3 pass per counter group)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  echo "== group $i: $grp" | tee -a gpurun_out/pmc/log.txt
  CD_ZS_DBG=${ZS_DBG:-0} timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc/g$i -o pmc -- python3 tools/conv_bench.py --iters 5 >> gpurun_out/pmc/log.txt 2>&1
  echo "rc=$?" | tee -a gpurun_out/pmc/log.txt
done
python3 - <<'PY'
import csv,glob,collections
for g in sorted(glob.glob('gpurun_out/pmc/g*')):
    for f in glob.glob(g+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'zslide' in r.get('Kernel_Name',''):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(g.split('/')[-1], k, 'n=%d mean=%.4g'%(len(v), sum(v)/len(v)))
PY
