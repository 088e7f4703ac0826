#!/bin/bash
# bash tools/train_prof.sh <tag>: the training bench line (batch 32, with its roofline leg) + per-kernel totals of the TIMED steps only
# (rocprofv3 kernel trace cut at the first timed step: tools/trace_stats.py), so that autotune probes and warm-up steps are outside.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-trainprof}; mkdir -p $out
bash tools/logrun.sh $out/train_bench.log python bench.py --mode train --steps 10 --warmup 3; tail -2 $out/train_bench.log | head -1 > $out/train_bench.json; cut -c1-300 $out/train_bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o bench -- python3 bench.py --mode train --steps 3 --warmup 3 --no-extra > $out/train_bench_under_rocprof.json 2> $out/tr.err; echo "trace rc=$?"
python3 tools/trace_stats.py $(find $out/tr -name "*kernel_trace.csv" | head -1) --marker pack_jobs_kernel --last 6 --steps 3 --out $out/train_kernel_stats.csv
rm -rf $out/tr
