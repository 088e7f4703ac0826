"""Inverse pre-processing of generated showers: ``utils.ReverseNorm`` of the reference (calodiffusion/utils/utils.py:253-257,
446-573) for the regular-grid datasets, on the device (``cd_reverse_norm``).

Supported: ``dataset_num`` 2 / 3, ``showerMap`` 'layer-logit-norm' / 'logit-norm' (the shipped Dataset-2 / Dataset-3 configs).
Quantile maps, Dataset-1 geometry conversion and the HGCal variant are not provided (they need files that do not ship)."""
import ctypes as C

import numpy as np
import torch

from . import engine

# normalisation constants of the reference (calodiffusion/utils/consts.py:82-116): data, not code
DATASET_PARAMS = {
    2: dict(logit_mean=-12.8564, logit_std=1.9123, totalE_mean=0.3926, totalE_std=0.05546, layers_mean=-6.35551, layers_std=3.90699),
    3: dict(logit_mean=-13.4753, logit_std=1.1070, totalE_mean=0.0, totalE_std=1.0, layers_mean=0.0, layers_std=1.0),
}


def ReverseNorm(voxels, e, hgcal=False, **kwargs):
    """Same call as the reference's ``utils.ReverseNorm``: returns (data (B, D*H*W) float32 ndarray, energy)."""
    if hgcal:
        raise NotImplementedError("ReverseNorm: the HGCal variant is not provided")
    return ReverseNormCaloChall(voxels, e, **kwargs)


def ReverseNormCaloChall(voxels, e, emax=9999.0, emin=0.0001, config=None, shape=None, binning_file="", max_deposit=2, logE=True,
                         layerE=None, showerMap="log", dataset_num=2, orig_shape=False, ecut=0.0, **kwargs):
    if dataset_num not in DATASET_PARAMS or orig_shape:
        raise NotImplementedError("ReverseNorm: only the regular-grid datasets 2 and 3 are provided")
    if showerMap not in ("layer-logit-norm", "logit-norm"):
        raise NotImplementedError("ReverseNorm: showerMap '%s' is not provided" % showerMap)
    c = DATASET_PARAMS[dataset_num]
    e = np.asarray(e, dtype=np.float32)
    energy = emin * (emax / emin) ** e if logE else emin + (emax - emin) * e   # utils.py:480-483, host numpy like the reference
    layer_mode = "layer" in showerMap
    if layer_mode and layerE is None:
        raise AssertionError("layerE is required for a 'layer' shower map")
    v = torch.as_tensor(voxels, dtype=torch.float32).cuda().contiguous()
    B, D, H, W = v.shape[0], v.shape[-3], v.shape[-2], v.shape[-1]
    en = torch.as_tensor(np.ascontiguousarray(energy.reshape(B), dtype=np.float32)).cuda()
    le = torch.as_tensor(np.asarray(layerE, dtype=np.float32)).cuda().contiguous() if layer_mode else None
    if le is not None and tuple(le.shape) != (B, D + 1):
        raise ValueError("ReverseNorm: layerE must have shape (batch, 1 + layers)")
    out = torch.empty((B, D * H * W), dtype=torch.float32, device="cuda")
    lib = engine.load_library()
    dims = (C.c_int32 * 3)(D, H, W)
    consts = (C.c_float * 6)(c["logit_mean"], c["logit_std"], c["totalE_mean"], c["totalE_std"], c["layers_mean"], c["layers_std"])
    engine._check(lib.cd_reverse_norm(v.data_ptr(), en.data_ptr(), le.data_ptr() if le is not None else None, out.data_ptr(), B,
                                      dims, consts, float(max_deposit), float(ecut), engine._stream()))
    return out.cpu().numpy(), energy
