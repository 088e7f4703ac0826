"""CPU, world_size 2 over gloo: the data-parallel sampling bookkeeping (batch sharding with no data-path collective,
max-over-ranks timing reduction used by bench.py).  The denoiser itself is stood in for by the oracle (tests only)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import seeded_unet


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from calodiffusion_amd import utils
    from calodiffusion_amd.configs import load_config
    from oracle import torch_oracle as O
    import bench

    cfg = load_config("tiny")
    m = O.OracleModel(cfg, seeded_unet("tiny").state_dict())
    g = torch.Generator().manual_seed(5)
    B = 6
    start = torch.randn((B, 1, 8, 8, 8), generator=g)
    E = torch.rand((B, 3), generator=g)
    layers = torch.randn((B, 9), generator=g)
    sl = utils.shard_batch(B, world, rank)
    x, _, _ = m.ddim_sample(start[sl], E[sl], layers[sl], 3)
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), x.numpy())
    # the timing reduction of bench.py: max over ranks
    t = bench.max_over_ranks(1.0 + rank)
    assert t == float(world), t
    if rank == 0:
        full, _, _ = m.ddim_sample(start, E, layers, 3)
        np.save(os.path.join(out_dir, "full.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from calodiffusion_amd.utils import allreduce_mean_
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(2212785, generator=g)  # Dataset-2 parameter count: the flat gradient buffer of one rank
    np.save(os.path.join(out_dir, f"g{rank}.npy"), flat.numpy())
    allreduce_mean_(flat)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_is_the_mean_over_ranks(tmp_path):
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    mean = sum(np.load(tmp_path / f"g{r}.npy") for r in range(world)) / world
    for r in range(world):
        assert np.allclose(np.load(tmp_path / f"r{r}.npy"), mean, rtol=0, atol=1e-7)
    # outside a process group it is the identity
    from calodiffusion_amd.utils import allreduce_mean_
    x = torch.arange(5.0)
    assert torch.equal(allreduce_mean_(x.clone()), x)


def test_sharded_sampling_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    parts = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)])
    # no op in the denoiser mixes batch elements => the union of shards IS the unsharded result
    # (torch's CPU kernels pick batch-size dependent algorithms: equality up to fp32 reorder noise)
    assert np.linalg.norm(parts - full) / np.linalg.norm(full) < 1e-5
