#!/usr/bin/env python3
"""Per-kernel totals of the TIMED region of a bench run from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`), in the
column layout of rocprofv3's own `--stats` file.  `--stats` covers the whole process: kernel autotuning probes, first-call packs and
the warm-up steps sit in its totals (the training profile mixed them in).  Here the region is cut at a marker kernel:
rows from the N-th last launch of `--marker` on (training: every step launches `pack_jobs_kernel` twice -- the weight re-pack and
the input-gradient images -- so `--marker pack_jobs_kernel --last 2*steps`).

    python3 tools/trace_stats.py <kernel_trace.csv> --marker pack_jobs_kernel --last 12 --out train_kernel_stats.csv
"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--marker", required=True)
ap.add_argument("--last", type=int, required=True, help="the region starts at the N-th last launch of the marker kernel")
ap.add_argument("--out", required=True)
ap.add_argument("--steps", type=int, default=1, help="steps in the region (printed per-step figures)")
ap.add_argument("--list", default=None, help="also print every launch of kernels whose name contains this (duration, grid, LDS) of the LAST step")
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
if len(marks) < a.last:
    raise SystemExit(f"only {len(marks)} launches of {a.marker!r} in the trace")
rows = rows[marks[-a.last]:]
agg = collections.OrderedDict()
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
    e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
tot = sum(e[1] for e in agg.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
with open(a.out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, e[0], e[1], f"{e[1] / e[0]:.1f}", f"{100.0 * e[1] / tot:.2f}", e[2], e[3]])
print(f"timed region: {len(rows)} launches, kernel time {tot / 1e6:.3f} ms, wall span {span / 1e6:.3f} ms over {a.steps} step(s): "
      f"{tot / 1e6 / a.steps:.3f} ms of kernels and {len(rows) / a.steps:.0f} launches per step")
for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {k[:100]:100s} calls/step {e[0] / a.steps:6.1f} avg {e[1] / e[0] / 1e3:8.1f} us  {100.0 * e[1] / tot:5.1f} %")
if a.list:
    per = len(rows) // a.steps
    for r in rows[-per:]:
        if a.list in r["Kernel_Name"]:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print(f"  {d / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')}x{r.get('Grid_Size_Z', '?')} wg {r.get('Workgroup_Size_X', '?')} "
                  f"lds {r.get('LDS_Block_Size', '?')}  {r['Kernel_Name'][:60]}")
