#!/usr/bin/env python3
"""HBM-side traffic of the level-0 HBM passes of one denoise step -- gn_apply, the two attention passes, the 1x1 shortcut conv, the
init conv and the head -- from rocprofv3 PMC counters, against the bytes each pass must move ("report
FETCH / WRITE PMC bytes against 53 / 106 / 159 MB").

    python3 tools/l0_traffic.py [--config dataset2 --batch 64]     # on the GPU box; writes gpurun_out/l0_traffic_<config>.json

One rocprofv3 pass per counter (FETCH_SIZE / WRITE_SIZE do not fit together), each running `python3 tools/step_breakdown.py`
(eager denoise steps).  Units and the gfx950 correction as in tools/zs_traffic.py (/opt/skills/guides/MI355X_MICROARCH.md): KiB per
dispatch, FETCH_SIZE doubled.  Only dispatches on the level-0 grid are kept: a kernel's launches are told apart by their grid size
(the level-0 launch of a kernel is the one with the most workgroups).  This script does not touch the GPU itself.
"""
import argparse
import collections
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, "gpurun_out", "pmc_l0")
KERNELS = {  # name fragment -> (label, algorithmic tensors read, written; in units of one (B, vox, 32) fp32 tensor)
    "gn_apply_kernel": ("gn_apply (GroupNorm + SiLU + shortcut)", 2, 1),
    "attn_kv_context_kernel": ("attention pass 1 (k, v, context)", 1, 0),
    "attn_out_kernel": ("attention pass 2 (q, output)", 1, 1),
    "pointwise_kernel": ("1x1 shortcut conv 64->32 closing its block", 3, 1),
    "head_gn_kernel": ("head (final block close + 32->1 conv + update)", 2, 0),
    "init_conv_f16x2_kernel": ("init conv (1 -> 32 channels)", 0, 1),
}


def read_pass(root):
    """{kernel fragment: {grid size: [counter values]}} and durations likewise."""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/**/*_results.db", recursive=True)):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        T = lambda key: next(t for t in tabs if t.startswith("rocpd_" + key))  # noqa: E731
        ks = {r[0]: r[1] for r in con.execute(f"select id, kernel_name from {T('info_kernel_symbol')}")}
        cols = [r[1] for r in con.execute(f"pragma table_info({T('kernel_dispatch')})")]
        gx = [c for c in ("grid_size_x", "grid_x") if c in cols][0]
        gy = [c for c in ("grid_size_y", "grid_y") if c in cols][0]
        gz = [c for c in ("grid_size_z", "grid_z") if c in cols][0]
        disp = {r[0]: r[1:] for r in con.execute(f"select event_id, kernel_id, start, end, {gx}, {gy}, {gz} from {T('kernel_dispatch')}")}
        for ev, pid, val in con.execute(f"select event_id, pmc_id, value from {T('pmc_event')}"):
            if ev not in disp:
                continue
            kid, st, en, x, y, z = disp[ev]
            name = ks.get(kid, "")
            for frag in KERNELS:
                if frag in name:
                    vals[frag][x * y * z].append(val)
        for ev, (kid, st, en, x, y, z) in disp.items():
            name = ks.get(kid, "")
            for frag in KERNELS:
                if frag in name:
                    durs[frag][x * y * z].append(en - st)
    return vals, durs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="dataset2")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--vox", type=int, default=6480)
    a = ap.parse_args()
    os.makedirs(SCRATCH, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    tensor = a.batch * a.vox * 32 * 4.0
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(SCRATCH, counter)
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--", "python3",
               os.path.join(ROOT, "tools", "step_breakdown.py"), "--config", a.config, "--batch", str(a.batch), "--reps", "3"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=500)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:], file=sys.stderr)
            raise SystemExit(f"rocprofv3 pass for {counter} failed ({r.returncode})")
        vals, durs = read_pass(d)
        for frag in KERNELS:
            if not vals[frag]:
                continue
            g = max(vals[frag])  # the level-0 launches: the largest grid of this kernel
            v, t = vals[frag][g], durs[frag][g]
            per.setdefault(frag, {})[counter] = {"kib_per_dispatch": sum(v) / len(v), "dispatches": len(v), "mean_us": sum(t) / len(t) / 1e3,
                                                "grid_threads": g}
        subprocess.run(["rm", "-rf", d])
    out = {"config": a.config, "batch": a.batch, "level0_voxels": a.vox, "tensor_bytes": tensor,
           "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) --kernel-trace -- python3 tools/step_breakdown.py; KiB per "
                     "dispatch, FETCH_SIZE x2 on gfx950; level-0 launches = the largest grid of each kernel", "kernels": {}}
    for frag, (label, nr, nw) in KERNELS.items():
        if frag not in per or "FETCH_SIZE" not in per[frag] or "WRITE_SIZE" not in per[frag]:
            continue
        fb = per[frag]["FETCH_SIZE"]["kib_per_dispatch"] * 1024.0 * 2.0
        wb = per[frag]["WRITE_SIZE"]["kib_per_dispatch"] * 1024.0
        us = per[frag]["FETCH_SIZE"]["mean_us"]
        out["kernels"][frag] = {"what": label, "fetch_MB": round(fb / 1e6, 1), "write_MB": round(wb / 1e6, 1),
                                "algorithmic_read_MB": round(nr * tensor / 1e6, 1), "algorithmic_write_MB": round(nw * tensor / 1e6, 1),
                                "mean_us_under_pmc": round(us, 1), "hbm_TBps": round((fb + wb) / (us * 1e-6) / 1e12, 2),
                                "dispatches": per[frag]["FETCH_SIZE"]["dispatches"]}
        print(f"{frag:28s} fetch {fb / 1e6:7.1f} MB (alg {nr * tensor / 1e6:6.1f})  write {wb / 1e6:7.1f} MB (alg {nw * tensor / 1e6:6.1f})  "
              f"{us:6.1f} us  {(fb + wb) / (us * 1e-6) / 1e12:5.2f} TB/s")
    path = os.path.join(ROOT, "gpurun_out", f"l0_traffic_{a.config}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
