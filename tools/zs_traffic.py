#!/usr/bin/env python3
"""HBM-side traffic of the dominant kernel (the z-slide 3x3x3 convolution at Dataset-2 level 0, batch 64) from rocprofv3 PMC
counters, written to profiles/zslide_traffic.json, which bench.py attaches to its `roofline.traffic` field.

    python3 tools/zs_traffic.py            # on the GPU box; one rocprofv3 pass per counter (FETCH_SIZE / WRITE_SIZE do not fit
                                           # together), each launching `python3 tools/conv_bench.py --iters 5` directly

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KiB, and FETCH_SIZE is doubled on gfx950 for 16-B/lane coalesced reads.  The JSON is
stamped with the sha256 of the kernel source: bench.py only reports the figure while the kernel is the one that was measured.
This script does not touch the GPU itself (it only spawns rocprofv3), so the profiler's preload rule is respected.
"""
import collections
import glob
import hashlib
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "calodiffusion_amd", "csrc", "kernels_conv_zs.hip")
OUT = os.path.join(ROOT, "profiles", "zslide_traffic.json")
SCRATCH = os.path.join(ROOT, "gpurun_out", "pmc_traffic")


def kernel_source_hash() -> str:
    return hashlib.sha256(open(SRC, "rb").read()).hexdigest()


def read_pass(root, pattern):
    """(mean counter value per dispatch, dispatch count, mean duration ns) of kernels whose name contains `pattern`."""
    vals, durs = collections.defaultdict(list), []
    for f in sorted(glob.glob(root + "/**/*_results.db", recursive=True)):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        T = lambda key: next(t for t in tabs if t.startswith("rocpd_" + key))  # noqa: E731
        ks = {r[0]: r[1] for r in con.execute(f"select id, kernel_name from {T('info_kernel_symbol')}")}
        pm = {r[0]: r[1] for r in con.execute(f"select id, name from {T('info_pmc')}")}
        disp = {r[0]: (r[1], r[2], r[3]) for r in con.execute(f"select event_id, kernel_id, start, end from {T('kernel_dispatch')}")}
        for ev, pid, val in con.execute(f"select event_id, pmc_id, value from {T('pmc_event')}"):
            if ev in disp and pattern in ks.get(disp[ev][0], ""):
                vals[pm[pid]].append(val)
        for ev, (kid, st, en) in disp.items():
            if pattern in ks.get(kid, ""):
                durs.append(en - st)
    return {k: sum(v) / len(v) for k, v in vals.items()}, len(durs), (sum(durs) / max(1, len(durs)))


CASES = {  # BASELINE.json's sampling configurations: level-0 grid, batch per GPU, output file
    "dataset2": ((45, 16, 9), 64, "zslide_traffic.json"),           # whole planes
    "dataset3": ((45, 50, 18), 32, "zslide_traffic_dataset3.json"),  # 10 phi strips of 5 rows (+ halo rows)
    "hgcal": ((28, 12, 21), 16, "zslide_traffic_hgcal.json"),        # 3 phi strips of 4 rows
}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="dataset2", choices=sorted(CASES))
    a = ap.parse_args()
    dims, batch, fname = CASES[a.config]
    global OUT
    OUT = os.path.join(ROOT, "profiles", fname)
    os.makedirs(SCRATCH, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(SCRATCH, counter)
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", "python3", os.path.join(ROOT, "tools", "conv_bench.py"),
               "--iters", "5", "--dims", ",".join(str(v) for v in dims), "--batch", str(batch)]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=400)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:], file=sys.stderr)
            raise SystemExit(f"rocprofv3 pass for {counter} failed ({r.returncode})")
        vals, n, dur = read_pass(d, "zslide")
        res[counter] = {"mean_per_dispatch": vals.get(counter), "dispatches": n, "mean_ns": dur}
        print(counter, res[counter])
    fetch_kib, write_kib = res["FETCH_SIZE"]["mean_per_dispatch"], res["WRITE_SIZE"]["mean_per_dispatch"]
    fetch_b = fetch_kib * 1024.0 * 2.0  # gfx950: doubled for 16-B/lane coalesced reads (MI355X_MICROARCH.md)
    write_b = write_kib * 1024.0
    vox = dims[0] * dims[1] * dims[2]
    out = {"kernel": "conv3x3x3_s1 C32->32 @%dx%dx%d" % dims, "batch": batch, "kernel_source_sha256": kernel_source_hash(),
           "fetch_bytes": fetch_b, "write_bytes": write_b, "traffic_bytes": fetch_b + write_b,
           "algorithmic_bytes": 2.0 * batch * vox * 32 * 4, "raw": res,
           "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --kernel-trace -- python3 tools/conv_bench.py --iters 5 "
                     "--dims ... --batch ...; KiB per dispatch; FETCH_SIZE x2 on gfx950"}
    for path in (OUT, os.path.join(ROOT, "gpurun_out", fname)):  # (gpurun merges gpurun_out/ back, not profiles/)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
    print("wrote", OUT, f"traffic {out['traffic_bytes'] / 1e6:.1f} MB vs algorithmic {out['algorithmic_bytes'] / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
