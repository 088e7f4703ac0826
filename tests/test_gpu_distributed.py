"""GPU (MI355X), two ranks on the one leased GPU: the data-parallel PRODUCT path end to end -- `compute_loss(...).backward()`
through `_TrainStep` with the flat-buffer gradient all-reduce on a device tensor, and batch-sharded sampling with per-rank rows
of one global Philox stream.  The ranks are forks of a GPU-free fork server (tests/conftest.py) and initialise the GPU
themselves; the process group is gloo (RCCL needs one GPU per rank), so `allreduce_mean_` bounces the device buffer through the
host -- the reduction arithmetic and every other line of the path are the ones an 8-GPU RCCL job runs."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORLD = 2
B_GLOBAL = 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(rank):
    g = torch.Generator().manual_seed(100 + rank)
    data = torch.randn((3, 1, 8, 8, 8), generator=g)
    E, layers = torch.rand((3, 3), generator=g), torch.randn((3, 9), generator=g)
    noise, rnd = torch.randn(data.shape, generator=g), torch.randn((3,), generator=g)
    return data, E, layers, noise, rnd


def _global_conditions():
    g = torch.Generator().manual_seed(77)
    return torch.rand((B_GLOBAL, 3), generator=g), torch.randn((B_GLOBAL, 9), generator=g)


def _model():
    from calodiffusion_amd.calodiffusion import CaloDiffusion
    from calodiffusion_amd.configs import load_config
    cfg = load_config("tiny")
    torch.manual_seed(1234)
    return CaloDiffusion(cfg, n_steps=cfg["NSTEPS"], loss_type=cfg["LOSS_TYPE"])


def _grads(m, rank):
    data, E, layers, noise, rnd = (v.cuda() for v in _inputs(rank))
    m.zero_grad()
    loss = m.compute_loss(data, E, noise=noise, layers=layers, rnd_normal=rnd)
    loss.backward()
    return float(loss), np.concatenate([p.grad.detach().cpu().numpy().ravel() for p in m.parameters()])


def _rank_main(rank, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from calodiffusion_amd.utils import shard_batch
    m = _model()
    loss, g = _grads(m, rank)  # _TrainStep.forward all-reduces the flat device buffer over the group
    np.save(os.path.join(out_dir, f"grad{rank}.npy"), g)
    # batch-sharded sampling: this rank's rows of the global batch and of the global Philox stream (bench.py does the same)
    E, layers = _global_conditions()
    sl = shard_batch(B_GLOBAL, WORLD, rank)
    m.set_noise_shard(sl.start, B_GLOBAL)
    m.noise_offset = 0
    outs = [m.sample(E[sl].cuda(), layers[sl].cuda(), num_steps=6) for _ in range(2)]  # tiny config: DDPM (stochastic)
    np.save(os.path.join(out_dir, f"sample{rank}.npy"), np.stack(outs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_train_step_and_sharded_sampling(tmp_path):
    if mp.get_start_method(allow_none=True) != "forkserver":
        pytest.skip("fork server not running (tests/conftest.py starts it for -m gpu runs)")
    ctx = mp.get_context("forkserver")
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, port, str(tmp_path))) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("rank timed out")
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    # single-process references, in this process
    m = _model()
    singles = [_grads(m, r)[1] for r in range(WORLD)]
    mean = sum(singles) / WORLD
    for r in range(WORLD):
        got = np.load(tmp_path / f"grad{r}.npy")
        assert np.array_equal(got, np.load(tmp_path / "grad0.npy"))  # identical replicas after the all-reduce
        err = np.linalg.norm(got - mean) / np.linalg.norm(mean)
        assert err < 1e-6, (r, err)
    assert np.linalg.norm(singles[0] - singles[1]) / np.linalg.norm(mean) > 1e-2  # the ranks really had different data
    E, layers = _global_conditions()
    m.noise_offset = 0
    full = np.stack([m.sample(E.cuda(), layers.cuda(), num_steps=6) for _ in range(2)])
    parts = np.concatenate([np.load(tmp_path / f"sample{r}.npy") for r in range(WORLD)], axis=1)
    err = np.linalg.norm(parts - full) / np.linalg.norm(full)
    assert err < 2e-6, err  # same noise bits; kernel tilings (hence fp32 summation order of the statistics) depend on the batch
    assert not np.allclose(full[0], full[1])


def test_bench_goes_through_rccl_with_one_rank():
    """The RCCL branch of bench.py (init_process_group("nccl"), barrier, MAX all-reduce, and in train mode the flat-gradient SUM
    all-reduce of utils.allreduce_mean_) on the ONE GPU a test box has: one rank under torch.distributed.run with
    --force-collective.  The multi-GPU run is the driver's; this shows the calls are accepted by RCCL on this image and the line
    says which backend carried them."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "train", "--steps", "2", "--warmup", "1",
           "--no-extra", "--force-collective"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = line["collective"]
    print("one-rank RCCL:", c)
    assert c["backend"] == "nccl" and c["ranks"] == 1 and c["allreduces_per_step"] == 1.0 and c["allreduce_ms"] > 0
