#!/bin/bash
# bash tools/prof_mode.sh <tag> [env assignments and bench args...]: rocprofv3 kernel stats of one bench configuration, top 25 rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-clocks "$@" > $out/bench_under_rocprof.json 2> $out/stats.err; echo "stats rc=$?"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/stats
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", tot/1e6)
for r in rows[:25]:
    print(f"{r['Name'][:105]:105s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1000:8.1f}us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
cut -c1-300 $out/bench_under_rocprof.json
