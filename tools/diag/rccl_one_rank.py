"""RCCL on the single GPU of a gpurun box: one rank under torch.distributed.run, backend "nccl" (= RCCL on ROCm) -- communicator
creation, barrier, the flat-gradient SUM all-reduce of the training step (8.85 MB) and the MAX all-reduce bench.py times with.
The multi-GPU run itself is the driver's; this shows the calls this repo makes are accepted by RCCL on this image."""
import os, time, torch, torch.distributed as dist
lr = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
print("backend", dist.get_backend(), "world", dist.get_world_size(), flush=True)
dist.barrier()
flat = torch.arange(2216448, dtype=torch.float32, device="cuda")
want = flat.clone()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
assert torch.equal(flat, want)
t = torch.tensor([3.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 3.25
dist.barrier()
print(f"ok: all_reduce of {flat.numel() * 4 / 1e6:.2f} MB in {dt * 1e6:.1f} us (1 rank), MAX all-reduce, barrier", flush=True)
dist.destroy_process_group()
