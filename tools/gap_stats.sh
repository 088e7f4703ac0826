#!/bin/bash
# bash tools/gap_stats.sh <tag> [bench args]: idle time between consecutive kernels of the replayed step graph (rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o bench -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-clocks "$@" > $out/bench.json 2> $out/tr.err; echo "trace rc=$?"
f=$(find $out/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 400 x N kernels are the timed pass: take the final 20000 rows
rows = rows[-20000:]
gaps = collections.defaultdict(list)
tot_k = tot_g = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 50000: continue  # (host-side pauses between graph launches / passes)
    gaps[(a["Kernel_Name"][:50], b["Kernel_Name"][:50])].append(g)
    tot_g += g
    tot_k += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
print("kernel time %.3f ms, gaps %.3f ms (%.1f %%) over %d launches: mean gap %.2f us" % (tot_k / 1e6, tot_g / 1e6, 100.0 * tot_g / (tot_k + tot_g), len(rows), tot_g / 1e3 / len(rows)))
worst = sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]
for (a, b), v in worst:
    print("  %-50s -> %-50s n %5d mean %.2f us" % (a, b, len(v), sum(v) / len(v) / 1e3))
PY
rm -rf $out/tr
