"""Host-side helpers of the hot path (the tiny subset of reference calodiffusion/utils/utils.py it needs)."""
from __future__ import annotations

from typing import List, Literal, Sequence, Tuple

import numpy as np
import torch

from .configs import LoadJson  # noqa: F401  (same name as the reference helper)

# r-bin edges per DATASET_NUM (utils/utils.py:45-127); HGCal (>= 100) uses unit-width bins
_R_EDGES = {
    0: [0.0, 1.0, 4.0, 5.0, 7.0, 10.0, 15.0, 20.0, 30.0, 50.0, 80.0, 90.0, 100.0, 130.0, 150.0, 160.0, 200.0, 250.0, 300.0,
        350.0, 400.0, 600.0, 1000.0, 2000.0],
    1: [0.0, 2.0, 4.0, 5.0, 6.0, 8.0, 10.0, 12.0, 15.0, 20.0, 25.0, 30.0, 40.0, 50.0, 60.0, 70.0, 80.0, 90.0, 100.0, 120.0,
        130.0, 150.0, 160.0, 200.0, 250.0, 300.0, 350.0, 400.0, 600.0, 1000.0, 2000.0],
    # Dataset-2/3 edges are spelt out as decimal literals in the reference (i*4.65 differs in the last bit for some i)
    2: [0, 4.65, 9.3, 13.95, 18.6, 23.25, 27.9, 32.55, 37.2, 41.85],
    3: [0, 2.325, 4.65, 6.975, 9.3, 11.625, 13.95, 16.275, 18.6, 20.925, 23.25, 25.575, 27.9, 30.225, 32.55, 34.875,
        37.2, 39.525, 41.85],
}


def get_device() -> torch.device:
    """Same policy as the reference (utils/utils.py:1034-1039)."""
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def coordinate_profiles(dataset_num: int, shape_dhw: Sequence[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """1-D fp32 profiles (R over r, Z over z, phi over phi) of the constant input images.

    Equivalent to create_R_Z_image(scaled=True) / create_phi_image (utils/utils.py:33-150): those images vary along one
    axis only, so the HIP init-conv loader rebuilds them from these profiles instead of reading three full tensors.
    """
    nz, nphi, nr = (int(v) for v in shape_dhw)
    edges: List[float] = [float(i) for i in range(nr + 1)] if dataset_num >= 100 else list(_R_EDGES[dataset_num])
    centres = [(edges[i] + edges[i + 1]) / 2.0 for i in range(len(edges) - 1)]
    if len(centres) != nr:
        raise ValueError(f"Mismatch for dataset size {tuple(shape_dhw)} and dataset num {dataset_num} - expecting "
                         f"dataset with final dim {len(centres)}")
    r = (torch.tensor(centres, dtype=torch.float32) / centres[-1]).numpy()
    z = (torch.arange(nz, dtype=torch.float32) / nz).numpy()
    phi = torch.linspace(0.0, 1.0, nphi, dtype=torch.float32).numpy()
    return r, z, phi


def create_R_Z_image(device, dataset_num=1, scaled=True, shape=(1, 45, 16, 9)):
    """API-compatible with the reference helper; built from the 1-D profiles."""
    r, z, _ = coordinate_profiles(dataset_num, shape[-3:])
    R = torch.from_numpy(r).view(1, 1, 1, -1).expand(*shape).clone()
    Z = torch.from_numpy(z).view(1, -1, 1, 1).expand(*shape).clone()
    if not scaled:
        raise NotImplementedError("only the scaled images are used by the denoiser")
    return R.to(device), Z.to(device)


def create_phi_image(device, shape=(1, 45, 16, 9)):
    phi = torch.linspace(0.0, 1.0, shape[-2], dtype=torch.float32)
    return phi.view(1, 1, -1, 1).expand(*shape).clone().to(device)


def subsample_alphas(alpha: torch.Tensor, time: torch.Tensor, x_shape) -> torch.Tensor:
    """utils/utils.py:1041-1044."""
    out = alpha.gather(-1, time.cpu())
    return out.reshape(time.shape[0], *((1,) * (len(x_shape) - 1))).to(time.device)


def load_attr(type_: Literal["sampler", "loss"], algo_name: str):
    """Resolve a sampler / loss class by name, as the reference does (utils/utils.py:1047-1061)."""
    if type_ == "sampler":
        from . import sample as module
    else:
        from . import loss as module
    try:
        return getattr(module, algo_name)
    except AttributeError as e:
        raise ValueError("%s '%s' is not supported: %s" % (type_, algo_name, e))


def shard_batch(n: int, world_size: int, rank: int) -> slice:
    """Contiguous batch shard of rank `rank` (sampling needs no collective: showers are independent)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


# measurement hook (bench.py --mode train): a list to which every device-side all-reduce appends its (start, end) events
ALLREDUCE_EVENTS = None
# bench.py --force-collective: run the all-reduce on a one-rank group too (a single-GPU box exercising the RCCL branch)
FORCE_SINGLE_RANK_COLLECTIVE = False


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """Data-parallel gradient exchange: ONE sum all-reduce of the flat fp32 gradient buffer (8.85 MB for Dataset-2) over
    the default process group (RCCL over xGMI on GPUs, gloo in the CPU tests), then the mean.  No-op outside a
    multi-process job.  This is the only collective of the training path."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_SINGLE_RANK_COLLECTIVE):
        if flat.is_cuda and dist.get_backend() == "gloo":
            # gloo moves host memory: bounce the buffer (tests with several ranks on one GPU; RCCL reduces in place on the device)
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        elif flat.is_cuda and ALLREDUCE_EVENTS is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            b.record()
            ALLREDUCE_EVENTS.append((a, b))
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    return flat
