#!/usr/bin/env python3
"""Shader clock and socket power of the GPU, sampled every few milliseconds, while a kernel loop runs back to back.

    python tools/clock_trace.py [--seconds 2 --batch 64 --period-ms 5 --out gpurun_out/clock_trace.json]

Measures DIRECTLY what DESIGN.md section 4 had inferred from in-kernel s_memtime stamps and the CU-scaling table: which clock
the chip sustains under the z-slide convolution (the dominant kernel, one workgroup per CU, MFMA-heavy), and how much power it
draws, against the same quantities when idle and under an HBM-bound elementwise kernel.  Sources tried in order: the amdsmi
python module (gpu_metrics: current_gfxclk(s), socket power), then sysfs (hwmon freq1_input / power1_average).  A sampler thread
polls while the main thread keeps the stream full (launches are asynchronous; one synchronise per ~100 launches).

Phases:  idle (0.5 s)  ->  conv loop (--seconds)  ->  idle (0.3 s)  ->  elementwise loop (--seconds / 2)  ->  idle.
Output: JSON {source, samples: [[t_ms, sclk_mhz, power_w, phase]...], summary: {phase: {sclk_mhz_mean, ..., launch_us}}}.
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Sampler:
    def __init__(self, period_s):
        self.period = period_s
        self.samples = []
        self.extra = []
        self.phase = "idle0"
        self.stop = False
        self.source = None
        self.read = self._pick()
        self.t0 = time.perf_counter()
        self.th = threading.Thread(target=self._run, daemon=True)

    def _pick(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            h = hs[0]
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            keys = [k for k in ("current_gfxclk", "current_gfxclks", "average_gfxclk_frequency", "average_socket_power",
                                "current_socket_power", "temperature_hotspot", "throttle_status", "indep_throttle_status") if k in m]
            self.source = f"amdsmi gpu_metrics {keys}"

            def num(v):
                return v if isinstance(v, (int, float)) and 0 < v < 60000 else None

            def rd():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clk = m.get("current_gfxclks") or m.get("current_gfxclk") or m.get("average_gfxclk_frequency")
                if isinstance(clk, (list, tuple)):
                    v = [c for c in clk if isinstance(c, (int, float)) and 0 < c < 10000]
                    clk = sum(v) / len(v) if v else None
                pw = m.get("current_socket_power")
                if not isinstance(pw, (int, float)) or pw <= 0 or pw > 5000:
                    pw = m.get("average_socket_power")
                # the other clock domains and the throttle status, when the driver reports them (kept per sample in self.extra)
                self.extra_last = {k: num(m.get(k)) for k in ("current_uclk", "average_uclk_frequency", "current_socclk",
                                                              "average_socclk_frequency", "average_fclk_frequency",
                                                              "temperature_hotspot", "temperature_mem")}
                self.extra_last["throttle_status"] = m.get("throttle_status")
                return clk, pw
            rd()
            return rd
        except Exception as e:  # noqa: BLE001
            self.amdsmi_error = repr(e)
        hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        for d in hw:
            f, p = os.path.join(d, "freq1_input"), None
            for cand in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(d, cand)):
                    p = os.path.join(d, cand)
            if os.path.exists(f):
                self.source = f"sysfs {d}"

                def rd(f=f, p=p):
                    clk = int(open(f).read()) / 1e6
                    pw = int(open(p).read()) / 1e6 if p else None
                    return clk, pw
                try:
                    rd()
                    return rd
                except Exception:  # noqa: BLE001
                    continue
        self.source = "none"
        return lambda: (None, None)

    def _run(self):
        while not self.stop:
            t = time.perf_counter()
            try:
                clk, pw = self.read()
            except Exception:  # noqa: BLE001
                clk, pw = None, None
            self.samples.append([round((t - self.t0) * 1e3, 2), clk, pw, self.phase])
            if getattr(self, "extra_last", None):
                self.extra.append(dict(self.extra_last, phase=self.phase))
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                time.sleep(dt)


def phase_summary(smp, phase, drop_head=0.2):
    """Means of one phase's samples (the first `drop_head` of it dropped as ramp): clocks in MHz, power in W."""
    rows = [s for s in smp.samples if s[3] == phase]
    rows = rows[int(len(rows) * drop_head):]
    clk = [r[1] for r in rows if r[1]]
    pw = [r[2] for r in rows if r[2]]
    out = {"samples": len(rows), "sclk_mhz_mean": round(sum(clk) / len(clk), 1) if clk else None,
           "sclk_mhz_min": min(clk) if clk else None, "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None,
           "power_w_max": max(pw) if pw else None}
    ex = [e for e in smp.extra if e.get("phase") == phase]
    ex = ex[int(len(ex) * drop_head):]
    for k in ("current_uclk", "average_uclk_frequency", "current_socclk", "average_fclk_frequency", "temperature_hotspot", "temperature_mem"):
        v = [e[k] for e in ex if e.get(k)]
        if v:
            out[k + "_mean"] = round(sum(v) / len(v), 1)
    th = {str(e.get("throttle_status")) for e in ex}
    if th - {"None"}:
        out["throttle_status_seen"] = sorted(th)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--period-ms", type=float, default=5.0)
    ap.add_argument("--out", default="gpurun_out/clock_trace.json")
    a = ap.parse_args()
    from calodiffusion_amd.engine import Ops
    ops = Ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn((a.batch, 45, 16, 9, 32), generator=g).cuda()
    w = (torch.randn((32, 32, 3, 3, 3), generator=g) * 0.05).cuda()
    b = torch.randn((32,), generator=g).cuda()
    big = torch.randn((64 * 6480 * 32,), generator=g).cuda()
    for _ in range(3):
        ops.cyl_conv(x, w, b)
    torch.cuda.synchronize()
    smp = Sampler(a.period_ms * 1e-3)
    smp.th.start()
    summary = {}

    def loop(name, fn, seconds):
        torch.cuda.synchronize()
        smp.phase = name
        n, t0 = 0, time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(100):
                fn()
            n += 100
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        summary[name] = {"launches": n, "launch_us": round(e0.elapsed_time(e1) * 1e3 / n, 2)}

    def idle(name, s):
        torch.cuda.synchronize()
        smp.phase = name
        time.sleep(s)

    idle("idle0", 0.5)
    loop("zslide_conv", lambda: ops.cyl_conv(x, w, b), a.seconds)   # (each call also runs the op entry point's tiny weight pack)
    idle("idle1", 0.3)
    loop("elementwise", lambda: big.mul_(1.0000001), a.seconds / 2)
    idle("idle2", 0.3)
    smp.stop = True
    smp.th.join()
    for ph in {s[3] for s in smp.samples}:
        rows = [s for s in smp.samples if s[3] == ph]
        # drop the first 20 % of a phase (ramp)
        rows = rows[len(rows) // 5:]
        clk = [r[1] for r in rows if r[1]]
        pw = [r[2] for r in rows if r[2]]
        d = summary.setdefault(ph, {})
        d.update(samples=len(rows), sclk_mhz_mean=round(sum(clk) / len(clk), 1) if clk else None,
                 sclk_mhz_min=min(clk) if clk else None, sclk_mhz_max=max(clk) if clk else None,
                 power_w_mean=round(sum(pw) / len(pw), 1) if pw else None, power_w_max=max(pw) if pw else None)
    out = {"source": smp.source, "amdsmi_error": getattr(smp, "amdsmi_error", None), "period_ms": a.period_ms, "batch": a.batch,
           "summary": summary, "samples": smp.samples}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"))
    print(json.dumps({"source": smp.source, "summary": summary}, indent=1))


if __name__ == "__main__":
    main()
