#!/bin/bash
# bash tools/train_list.sh <tag> <kernel-name-part> [env K=V ...]: kernel trace of three timed training steps, per-kernel totals and every
# launch of the named kernel in the last step (duration, grid, LDS)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-trainlist}; pat="$2"; shift; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o bench -- python3 bench.py --mode train --steps 3 --warmup 3 --no-extra > $out/train_bench_under_rocprof.json 2> $out/tr.err; echo "trace rc=$?"
python3 tools/trace_stats.py $(find $out/tr -name "*kernel_trace.csv" | head -1) --marker pack_jobs_kernel --last 6 --steps 3 --out $out/train_kernel_stats.csv --list "$pat" | tee $out/list.txt
rm -rf $out/tr
