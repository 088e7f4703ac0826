#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: how much of the busy time has two or more kernels in flight, and the per-queue split."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
q = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    q[r.get("Queue_Id", "?")] += e - s
ev.sort()
depth = 0; last = ev[0][0]; t = collections.Counter()
for ts, d in ev:
    t[depth] += ts - last; last = ts; depth += d
busy = sum(v for k, v in t.items() if k > 0)
print("kernels", len(rows), "queues", dict(q))
for k in sorted(t):
    print(f"depth {k}: {t[k]/1e6:9.3f} ms ({100*t[k]/max(1,sum(t.values())):5.1f} %)")
