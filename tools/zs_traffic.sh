#!/bin/bash
# HBM-side traffic of the z-slide conv kernel (rocprofv3 PMC, one pass per counter: FETCH_SIZE and WRITE_SIZE do not fit together)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_traffic/$c -o pmc -- python3 tools/conv_bench.py --iters 5 >> gpurun_out/pmc_traffic/log.txt 2>&1
  echo "$c rc=$?"
done
python3 tools/pmc_read.py gpurun_out/pmc_traffic zslide
