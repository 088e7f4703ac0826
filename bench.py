#!/usr/bin/env python3
"""Throughput of the hot path: sampled showers/sec (Dataset-2, 400-step DDIM, batch 64 per GPU) on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]           # N = 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: Diffusion.sample() = Philox start noise + 400 denoise steps + the
final device->host copy of x, inputs (E, layers) already resident in HBM.  One process per GPU; the batch
dimension is sharded with no data-path collective (weak scaling: 64 showers per GPU).  Rank 0 prints ONE JSON line.

Extra legs, rank 0 at N = 1 only:
  * roofline: one eager denoise step with HIP events around every launch (cd_profile_*), pricing the dominant kernel
    (the 32->32 3x3x3 cylindrical conv at full resolution) in algorithmic TFLOP/s against the fp32 MFMA peak;
  * cpu_baseline: the oracle (oracle/torch_oracle.py, stock PyTorch CPU kernels, the reference's own arithmetic
    boundary) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-in MFMA = vector fp32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA


def measured_traffic(kernel: str, batch: int):
    """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (tools/zs_traffic.py writes
    profiles/zslide_traffic.json: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE).  PMC counters cannot be collected inside this
    process, so the committed figure is attached -- only while the kernel source is the one that was measured (sha256 stamp),
    and only for the profiled kernel and batch; otherwise None."""
    import glob
    import hashlib
    try:
        src = open(os.path.join(ROOT, "calodiffusion_amd", "csrc", "kernels_conv_zs.hip"), "rb").read()
    except OSError:
        return None
    stamp = hashlib.sha256(src).hexdigest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "zslide_traffic*.json"))):  # one record per configuration
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("kernel") == kernel and rec.get("batch") == batch and rec.get("kernel_source_sha256") == stamp:
            return rec["traffic_bytes"]
    return None

BF16X3_TERMS = 6                # bf16 MFMAs per fp32 product in the split-bf16 convolution
PEAK_HBM_GBS = 8000.0


def dist_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# per-configuration defaults = the cases BASELINE.json names (configs[1], [3], [4]; [2] is --mode train)
CONFIG_DEFAULTS = {
    "dataset2": {"batch": 64, "sample_steps": 400, "sampler": "DDim"},   # the headline: 400-step DDIM, batch 64 per GPU
    "dataset3": {"batch": 32, "sample_steps": 400, "sampler": None},     # config's own sampler (DDim)
    "hgcal": {"batch": 16, "sample_steps": 200, "sampler": None},        # config's own sampler (DDPM): 128 showers over 8 GPUs
}
TRAIN_BATCH = 32  # BASELINE configs[2]: global batch 256 over 8 GPUs


def launcher_command(argv, gpus: int, port: int):
    """The command a bare `python bench.py --gpus N ...` (no WORLD_SIZE in the environment) re-launches itself as: N fresh
    ranks, one per GPU, under torch.distributed.run.  Built and started BEFORE this process touches the GPU (a process that
    has initialised HIP must neither fork ranks nor exec)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(argv, gpus: int) -> int:
    """Parent of a multi-GPU run started without a launcher: start the ranks as a child process group, relay rank 0's JSON
    line, exit with the child's code.  Fails loudly if the node has fewer GPUs than asked for (device_count does not
    initialise the GPU on this stack)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < gpus:
        raise SystemExit(f"bench.py --gpus {gpus}: this node exposes {have} GPU(s); refusing to report a {gpus}-GPU number")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = launcher_command(argv, gpus, port)
    print("[bench] launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


FORCE_COLLECTIVE = False  # --force-collective: a one-rank run still goes through RCCL (what a single-GPU box can show of that branch)


def collective_info():
    """What the timing barrier / max-over-ranks (and, in --mode train, the gradient all-reduce) actually ran on."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
    return {"backend": None, "ranks": 1}


def max_over_ranks(value: float) -> float:
    """Slowest rank's time (MAX all-reduce); identity in a single-process run."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not FORCE_COLLECTIVE):
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def synthetic_inputs(cfg, batch, rank, world, device):
    """E ~ U(0,1), layers ~ N(0,1) in normalised space.  ONE global set of world * batch showers (fixed seed), of
    which this rank holds its contiguous shard: together with set_noise_shard the union of the ranks' outputs is the
    single-GPU result of the same seed."""
    from calodiffusion_amd.utils import shard_batch
    g = torch.Generator().manual_seed(1234)
    n_e = 3 if cfg.get("HGCAL", False) else 1
    sl = shard_batch(world * batch, world, rank)
    E = torch.rand((world * batch, n_e), generator=g)[sl].contiguous().to(device)
    layers = None
    if "layer" in cfg.get("SHOWERMAP", ""):
        layers = torch.randn((world * batch, 1 + cfg["SHAPE_FINAL"][2]), generator=g)[sl].contiguous().to(device)
    return E, layers, sl


def cpu_baseline(cfg, sample_steps, batch, timed=3):
    """The oracle on the host cores at the SAME batch as the GPU run: `timed` denoise steps after a warm-up, extrapolated to
    sample_steps steps per shower (every DDIM step costs the same U-Net forward).  The thread count is the best of a quick
    {8, 16, 32} probe (one denoise step each).  More is pointless and expensive to find out: the box's CPU share is 16 for one
    GPU, PyTorch's CPU convolutions stop scaling long before, and at all 256 visible cores one step took 64 s (r2b)."""
    from oracle import torch_oracle as O
    from calodiffusion_amd.unet import CondUnet, unet_kwargs_from_config
    state = torch.random.get_rng_state()
    torch.manual_seed(1234)  # the same seeded weights as the GPU model (torch default init under this seed)
    net = CondUnet(**unet_kwargs_from_config(cfg))
    torch.random.set_rng_state(state)
    model = O.OracleModel(cfg, net.state_dict())
    g = torch.Generator().manual_seed(7)
    x = torch.randn([batch] + list(cfg["SHAPE_PAD"][1:]), generator=g)
    n_e = 3 if cfg.get("HGCAL", False) else 1
    E = torch.rand((batch, n_e), generator=g)
    layers = torch.randn((batch, 1 + cfg["SHAPE_FINAL"][2]), generator=g) if "layer" in cfg.get("SHOWERMAP", "") else None
    sig = torch.full((batch,), 1.0)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    probe = {}
    with torch.no_grad():
        for th in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
            torch.set_num_threads(th)
            model.denoise(x, E, sig, layers)  # warm-up at this thread count
            t0 = time.perf_counter()
            model.denoise(x, E, sig, layers)
            probe[th] = time.perf_counter() - t0
        threads = min(probe, key=probe.get)
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        for _ in range(timed):
            model.denoise(x, E, sig, layers)
        dt = (time.perf_counter() - t0) / timed
    torch.set_num_threads(default_threads)
    return {"value": batch / (dt * sample_steps), "unit": "showers/s", "cores": threads, "kind": "port",
            "host_cpus": ncpu, "s_per_denoise_step": dt, "thread_probe_s_per_step": {str(k): round(v, 3) for k, v in probe.items()},
            "sample": f"{timed} denoise steps (oracle/torch_oracle.py, PyTorch CPU fp32) at batch {batch} on {threads} threads "
                      f"(best of a {sorted(probe)} probe) after a warm-up, extrapolated x{sample_steps} steps"}


def roofline_leg(model, cfg, batch, E, layers):
    """One eager denoise step with HIP events around every kernel launch; returns the dominant kernel's roofline entry
    and a per-category breakdown."""
    from calodiffusion_amd import engine
    shape = [batch] + list(cfg["SHAPE_PAD"][1:])
    x = engine.randn(shape, "cuda", seed=99)
    sig = torch.full((batch,), 1.0, device="cuda")
    for _ in range(2):
        model.denoise(x, E=E, sigma=sig, layers=layers)
    torch.cuda.synchronize()
    reps = 5
    engine.profile_begin()
    for _ in range(reps):
        model.denoise(x, E=E, sigma=sig, layers=layers)
    for _ in range(20):  # the profiler's own cost per launch: an event pair around a one-workgroup kernel, in the same session
        engine.randn([64], "cuda", seed=1)
    prof = engine.profile_end()
    empty = prof.pop("randn", None)
    event_overhead_ms = empty["ms"] / empty["launches"] if empty and empty["launches"] else None
    total_ms = sum(v["ms"] for v in prof.values()) / reps
    # dominant kernel: the category with the largest total time
    dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms = dom["ms"] / dom["launches"]
    achieved = dom["flops"] / (avg_ms * 1e-3) / 1e12
    breakdown = {k: {"ms_per_step": round(v["ms"] / reps, 4), "launches_per_step": v["launches"] // reps,
                     "tflops": round(v["flops"] / (v["ms"] / v["launches"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                     "gbs": round(v["bytes"] / (v["ms"] / v["launches"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    # The convs compute fp32-grade results on the 16-bit matrix pipe.  Default (f16x2): every fp32 operand is split into two
    # fp16 terms and a MAC block takes 3 fp16 MFMAs, so the roof of the algorithm that runs is fp16-dense / 3.
    # CD_CONV_PRECISION=bf16x3 selects the exact 3-term bf16 split (6 MFMAs, roof bf16-dense / 6), =f32 the f32-input MFMA
    # kernels (roof = the fp32 matrix peak).
    mode = engine.get_conv_precision()
    peak = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / BF16X3_TERMS}.get(mode, PEAK_BF16_MFMA_TFLOPS / 3)
    pipe = {"f32": "f32 MFMA (157.3 TFLOP/s)",
            "bf16x3": "bf16 MFMA, fp32 operands split exactly into 3 bf16 terms, 6 MFMAs per MAC (2500/6 TFLOP/s)"}.get(
        mode, "fp16 MFMA, fp32 operands split into 2 fp16 terms (22 bits), 3 MFMAs per MAC block (2500/3 TFLOP/s)")
    roof = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 3), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "traffic": measured_traffic(dom_name, batch) if mode == "f16x2" else None,
            "pipe": pipe,
            "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
            # all 256 CUs under matrix load run at ~1.97 GHz: 2.06 PFLOP/s of dense fp16 MFMA measured (tools/micro/mfma_chain.hip)
            "frac_of_sustained_clock_peak": round(achieved / (2060.0 / 3), 4) if mode == "f16x2" else None,
            "avg_launch_us": round(avg_ms * 1e3, 2),
            # what the same event pair measures around a one-workgroup kernel (dispatch + event cost, ~5 us): `achieved` / `frac`
            # keep it in (conservative); net of it the launch agrees with the rocprofv3 average inside the replayed graph
            "event_pair_around_empty_kernel_us": round(event_overhead_ms * 1e3, 2) if event_overhead_ms else None,
            "frac_net_of_event_overhead": (round(dom["flops"] / ((avg_ms - event_overhead_ms) * 1e-3) / 1e12 / peak, 4)
                                           if event_overhead_ms and avg_ms > event_overhead_ms else None),
            "alg_flops_per_launch": dom["flops"],
            "alg_bytes_per_launch": dom["bytes"],
            "hbm_frac_of_same_kernel": round(dom["bytes"] / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "eager_step_ms_sum_of_kernels": round(total_ms, 3)}
    return roof, breakdown


def train_roofline_leg(model, step, data, E, layers, B, step_ms):
    """Roofline entry of the training step (outside the timed region).  Algorithmic work = 3 x the forward's FLOPs (forward, input
    gradients, weight gradients: every convolution / linear map runs three times), the forward's FLOPs being what the launchers of
    one eager denoise call at this batch declare (the SURVEY 8d accounting: 5.44 GFLOP per Dataset-2 shower); achieved =
    that / the measured step time, against the roof of the pipe the convolutions run on (fp16 MFMA / 3 MFMAs per MAC block).  Next
    to it the step's HBM-bound part: the GroupNorm backward passes (statistics + apply: dy and h read twice, dh written) with
    HIP events around their launches, as GB/s against the HBM roof."""
    from calodiffusion_amd import engine
    sig = torch.full((B,), 1.0, device="cuda")
    with torch.no_grad():
        model.denoise(data, E=E, sigma=sig, layers=layers)
        torch.cuda.synchronize()
        engine.profile_begin()
        model.denoise(data, E=E, sigma=sig, layers=layers)
        fwd = engine.profile_end()
    fwd_flops = sum(v["flops_total"] for v in fwd.values())
    engine.profile_begin()
    step()
    prof = engine.profile_end()
    gn = prof.get("gn_backward")
    mode = engine.get_conv_precision()
    peak = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / BF16X3_TERMS}.get(mode, PEAK_BF16_MFMA_TFLOPS / 3)
    achieved = 3.0 * fwd_flops / (step_ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": "whole training step (forward + input gradients + weight gradients)",
           "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           "traffic": None, "alg_flops_per_step": 3.0 * fwd_flops, "forward_gflop_per_sample": round(fwd_flops / B / 1e9, 3),
           "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
           "eager_step_ms_sum_of_profiled_launches": round(sum(v["ms"] for v in prof.values()), 3)}
    if gn and gn["launches"]:
        gbs = gn["bytes_total"] / (gn["ms"] * 1e-3) / 1e9
        out["groupnorm_backward"] = {"bound": "hbm", "launch_groups_per_step": gn["launches"], "ms_per_step": round(gn["ms"], 3),
                                     "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
    top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]
    out["top_categories_ms"] = {k: round(v["ms"], 3) for k, v in top}
    return out


def train_bench(args, model, cfg, E, layers, rank, world):
    """Training throughput (BASELINE configs[2]): zero_grad -> compute_loss -> backward -> Adam.step per iteration, as
    TrainDiffusion.training_loop does; data-parallel replicas with one flat gradient all-reduce per step."""
    B = args.batch
    from calodiffusion_amd import utils as cd_utils
    g = torch.Generator().manual_seed(4321 + rank)
    shape = [B] + list(cfg["SHAPE_PAD"][1:])
    data = torch.randn(shape, generator=g).cuda()
    noise = torch.randn(shape, generator=g).cuda()
    rnd = torch.randn((B,), generator=g).cuda()
    from calodiffusion_amd.optim import FusedAdam  # torch.optim.Adam semantics, one launch per 48 tensors (cd_adam_step)
    opt = (torch.optim.Adam if os.environ.get("CD_TORCH_ADAM") else FusedAdam)(model.parameters(), lr=float(cfg["LR"]))

    def step():
        opt.zero_grad()
        loss = model.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    cd_utils.ALLREDUCE_EVENTS = []  # (start, end) device events around every gradient all-reduce of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    events, cd_utils.ALLREDUCE_EVENTS = cd_utils.ALLREDUCE_EVENTS, None
    ar_ms = max_over_ranks(sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps))
    nbytes = model.engine().grad_layout()[1] * 4
    step_ms = 1e3 * dt / args.steps
    roof = None
    if rank == 0 and world == 1 and not args.no_extra:
        roof = train_roofline_leg(model, step, data, E, layers, B, step_ms)
    result = {"metric": f"training samples/sec ({args.config}, hybrid_weight l2, Adam)", "value": world * args.steps * B / dt,
              "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": f"{args.config} training step, batch {B} per GPU", "global_batch": B * world,
                         "parallelism": f"data-parallel x{world}, one flat fp32 gradient all-reduce per step",
                         "final_loss": float(loss)},
              "collective": dict(collective_info(), allreduce_bytes=nbytes, allreduces_per_step=len(events) / max(1, args.steps),
                                 allreduce_ms=ar_ms, allreduce_share_of_step=ar_ms / (1e3 * dt / args.steps))}
    if roof:
        result["roofline"] = roof
    if rank == 0:
        print(json.dumps(result))
    if world > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="dataset2")
    ap.add_argument("--batch", type=int, default=None,
                    help="showers per GPU (weak scaling); default: BASELINE's case of the config (64 / 32 / 16; 32 for --mode train)")
    ap.add_argument("--sample-steps", type=int, default=None, help="default: BASELINE's case of the config (400; hgcal 200)")
    ap.add_argument("--sampler", default=None, help="sampler class name; default: DDim for dataset2, else the config's own")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the roofline and cpu_baseline legs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel breakdown to stderr")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg only")
    ap.add_argument("--force-collective", action="store_true",
                    help="under a launcher with ONE rank: initialise RCCL anyway and run the barrier / MAX / gradient all-reduce through it")
    ap.add_argument("--no-clocks", action="store_true", help="do not sample the GPU's clocks / power during the timed region")
    ap.add_argument("--mode", default="sample", choices=["sample", "train"],
                    help="train: BASELINE configs[2], a step = one training iteration (fwd + bwd + grad all-reduce + Adam)")
    args = ap.parse_args()
    dflt = CONFIG_DEFAULTS.get(args.config, {"batch": 64, "sample_steps": 400, "sampler": None})
    if args.batch is None:
        args.batch = TRAIN_BATCH if args.mode == "train" else dflt["batch"]
    if args.sample_steps is None:
        args.sample_steps = dflt["sample_steps"]

    rank, local_rank, world = dist_info()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started without a launcher: become the parent of N fresh ranks (nothing has touched the GPU yet)
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for another GPU count")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if args.force_collective and "RANK" not in os.environ:
            raise SystemExit("--force-collective needs a launcher (python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 ...)")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if args.force_collective:
            global FORCE_COLLECTIVE
            FORCE_COLLECTIVE = True
            from calodiffusion_amd import utils as _u
            _u.FORCE_SINGLE_RANK_COLLECTIVE = True

    from calodiffusion_amd import engine as engine_mod
    from calodiffusion_amd.calodiffusion import CaloDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = dict(load_config(args.config))
    cfg["_name"] = args.config
    if args.sampler or dflt["sampler"]:
        cfg["SAMPLER"] = args.sampler or dflt["sampler"]
    sampler_name = cfg["SAMPLER"]
    cfg["SAMPLER_OPTIONS"] = dict(cfg.get("SAMPLER_OPTIONS") or {}, HIP_GRAPH=not args.no_graph)
    torch.manual_seed(1234)  # identical random-init weights on every rank
    model = CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])
    B = args.batch
    E, layers, shard = synthetic_inputs(cfg, B, rank, world, "cuda")
    # every rank walks the one global Philox stream and draws its own rows of each tensor (weak scaling: world * B showers)
    model.set_noise_shard(shard.start, world * B)

    if args.mode == "train":
        return train_bench(args, model, cfg, E, layers, rank, world)

    def one_pass():
        out = model.sample(E, layers, num_steps=args.sample_steps)  # returns a host ndarray (final D2H included)
        return out

    for _ in range(args.warmup):
        one_pass()
    # shader clock / socket power of this rank's GPU during the timed region (amdsmi gpu_metrics, sampled every 10 ms by a thread
    # that sleeps in between): the boxes of the pool differ by a few per cent at the same code, and the chip's power management
    # decides how fast the MFMA-heavy kernels run -- the line says under which conditions its number was measured
    smp = None
    if rank == 0 and not args.no_clocks:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import clock_trace
            smp = clock_trace.Sampler(0.010)
            smp.phase = "timed"
            smp.th.start()
        except Exception as e:  # noqa: BLE001 -- measurement garnish only
            print(f"[bench] clock sampling unavailable: {e!r}", file=sys.stderr)
            smp = None
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    gpu_state = None
    if smp is not None:
        smp.stop = True
        smp.th.join()
        gpu_state = dict(clock_trace.phase_summary(smp, "timed", drop_head=0.1), source=smp.source)
    assert np.isfinite(out).all()

    result = {
        "metric": "sampled showers/sec (Dataset-2, 400-step DDIM)"
        if args.config == "dataset2" and args.sample_steps == 400 and sampler_name == "DDim"
        else f"sampled showers/sec ({args.config}, {args.sample_steps}-step {sampler_name})",
        "value": world * args.steps * B / dt,
        "unit": "showers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (convs: fp32 operands as 2-term fp16 splits on the fp16 MFMA pipe, fp32 accumulate; attention/norms fp32)"
        if engine_mod.get_conv_precision() == "f16x2" else f"f32 (convs: {engine_mod.get_conv_precision()})",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {'x'.join(str(v) for v in cfg['SHAPE_PAD'][2:])} voxels, "
                               f"{args.sample_steps}-step {sampler_name}, batch {B} per GPU, random-init weights (seed 1234)",
                   "global_batch": B * world, "parallelism": f"batch-sharded x{world}, no collective",
                   "hip_graph": not args.no_graph, "denoise_ms": 1e3 * dt / args.steps / args.sample_steps},
        "collective": collective_info(),  # timing barrier + max-over-ranks only: the sampling path has no data-path collective
    }
    if gpu_state:
        result["gpu_state"] = gpu_state
    if rank == 0 and world == 1 and not args.no_extra:
        roof, breakdown = roofline_leg(model, cfg, B, E, layers)
        result["roofline"] = roof
        if args.breakdown:
            print(json.dumps(breakdown, indent=1), file=sys.stderr)
        result["kernel_breakdown_ms_per_denoise"] = {k: v["ms_per_step"] for k, v in list(breakdown.items())[:8]}
        if engine_mod.get_conv_precision() == "f16x2":
            # the exact-24-bit arithmetic (three-term bf16 split, 6 MFMAs per block) on record beside the headline
            engine_mod.set_conv_precision("bf16x3")
            one_pass()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one_pass()
            torch.cuda.synchronize()
            roof["bf16x3_showers_per_s"] = round(B / (time.perf_counter() - t0), 2)
            engine_mod.set_conv_precision("f16x2")
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(cfg, args.sample_steps, B)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1 or FORCE_COLLECTIVE:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
