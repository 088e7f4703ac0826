#!/bin/bash
# bash tools/train_prof.sh <tag>: training bench line (batch 32) + rocprofv3 kernel stats of the same command with the autotune
# probes and first-call work outside the trace window as far as --warmup allows (3 warm-up steps, 3 timed).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-trainprof}; mkdir -p $out
bash tools/logrun.sh $out/train_bench.log python bench.py --mode train --steps 10 --warmup 3; tail -3 $out/train_bench.log | cut -c1-400
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --mode train --steps 3 --warmup 3 > $out/train_bench_under_rocprof.json 2> $out/stats.err; echo "stats rc=$?"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/train_kernel_stats.csv; rm -rf $out/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms over 6 steps", tot/1e6)
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f}us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
