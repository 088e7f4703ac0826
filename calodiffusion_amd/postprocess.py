"""Inverse pre-processing of generated showers: ``utils.ReverseNorm`` of the reference (calodiffusion/utils/utils.py:253-257,
446-573) for the regular-grid datasets, on the device (``cd_reverse_norm``).

Supported: ``dataset_num`` 2 / 3, ``showerMap`` 'layer-logit-norm' / 'logit-norm' (the shipped Dataset-2 / Dataset-3 configs), and
the HGCal variant ``ReverseNormHGCal`` (utils/HGCal_utils.py:167-292) as two device stages around its geometry decode: the decoder
(``NN_embed`` with the reference's ``dec_batches``) is the caller's -- the geometry file it needs does not ship with the reference.
Quantile maps and the Dataset-1 geometry conversion are not provided."""
import ctypes as C

import numpy as np
import torch

from . import engine

# normalisation constants of the reference (calodiffusion/utils/consts.py:82-116): data, not code
DATASET_PARAMS = {
    2: dict(logit_mean=-12.8564, logit_std=1.9123, totalE_mean=0.3926, totalE_std=0.05546, layers_mean=-6.35551, layers_std=3.90699),
    3: dict(logit_mean=-13.4753, logit_std=1.1070, totalE_mean=0.0, totalE_std=1.0, layers_mean=0.0, layers_std=1.0),
}


# HGCal sets (consts.py:118-181)
DATASET_PARAMS.update({
    100: dict(logit_mean=-13.7371, logit_std=0.68639, totalE_mean=0.0055, totalE_std=0.00018, layers_mean=-4.4450, layers_std=2.37667),
    101: dict(logit_mean=-18.3170, logit_std=1.03153, totalE_mean=0.5538, totalE_std=0.01767, layers_mean=-4.5836, layers_std=2.98382),
    111: dict(logit_mean=-17.3442, logit_std=3.26085, totalE_mean=1.1076, totalE_std=0.03535, layers_mean=-4.5836, layers_std=2.98382),
    120: dict(logit_mean=-18.1561, logit_std=1.56255, totalE_mean=0.5389, totalE_std=0.30325, layers_mean=-6.7899, layers_std=5.64943),
    121: dict(logit_mean=-17.8664, logit_std=2.34207, totalE_mean=1.0270, totalE_std=0.09394, layers_mean=-11.6495, layers_std=7.31088),
})


def ReverseNorm(voxels, e, hgcal=False, **kwargs):
    """Same call as the reference's ``utils.ReverseNorm``: returns (data float32 ndarray, energy)."""
    if hgcal:
        return ReverseNormHGCal(voxels, e, **kwargs)
    return ReverseNormCaloChall(voxels, e, **kwargs)


def _staged(v, energy, layerE, dims, c, max_deposit, alpha, layer_eps, stage):
    lib = engine.load_library()
    B = v.shape[0]
    out = torch.empty((B, int(np.prod(dims))), dtype=torch.float32, device="cuda")
    consts = (C.c_float * 6)(c["logit_mean"], c["logit_std"], c["totalE_mean"], c["totalE_std"], c["layers_mean"], c["layers_std"])
    engine._check(lib.cd_reverse_norm_staged(v.data_ptr(), energy.data_ptr() if energy is not None else None,
                                             layerE.data_ptr() if layerE is not None else None, out.data_ptr(), B,
                                             (C.c_int32 * 3)(*dims), consts, float(max_deposit), 0.0, float(alpha), float(layer_eps),
                                             int(stage), engine._stream()))
    return out


def ReverseNormHGCal(voxels, e, shape=None, emax=9999.0, emin=0.0001, max_deposit=2, logE=True, layerE=None, showerMap="log",
                     dataset_num=2, orig_shape=False, ecut=0.0, embed=False, NN_embed=None, binning_file="", config=None,
                     sparse_decoding=False, sparse_per_batch=False):
    """``utils.ReverseNormHGCal`` (calodiffusion/utils/HGCal_utils.py:167-292), same arguments and return values.  The incident
    energy is LINEAR in e here (``emin + (emax - emin) e``, :195-199, whatever ``logE`` says), reverse_logit uses alpha 1e-8, the
    energy cut is disabled in the reference (``if ecut > 0 and False``).  ``embed``: the caller's ``NN_embed.dec_batches`` decodes
    between the two device stages (the reference builds an ``HGCalConverter`` from ``binning_file`` when none is given: that needs
    its geometry pickle, so here the converter must be passed in)."""
    if dataset_num not in DATASET_PARAMS:
        raise NotImplementedError("ReverseNormHGCal: no constants for dataset_num %r" % (dataset_num,))
    if "logit" not in showerMap or "norm" not in showerMap or "quantile" in showerMap:
        raise NotImplementedError("ReverseNormHGCal: showerMap '%s' is not provided ([layer-]logit-norm only)" % showerMap)
    c = DATASET_PARAMS[dataset_num]
    e = np.asarray(e, dtype=np.float32)
    gen_out = np.array(emin) + (np.array(emax) - np.array(emin)) * e
    energy = gen_out[:, 0]
    v = torch.as_tensor(voxels, dtype=torch.float32).cuda().contiguous()
    B = v.shape[0]
    data = _staged(v, None, None, (int(np.prod(v.shape[1:])), 1, 1), c, max_deposit, 1e-8, 1e-8, 1).reshape(v.shape)
    if embed:
        if NN_embed is None:
            raise NotImplementedError("ReverseNormHGCal(embed=True) needs the geometry converter (NN_embed): building one takes the "
                                      "geometry file of the HGCalShowers package, which does not ship with the reference")
        dec = NN_embed.dec_batches(data.cpu().numpy(), sparse_decoding=sparse_decoding, sparse_per_batch=sparse_per_batch)
        data = torch.as_tensor(np.asarray(dec, dtype=np.float32)).cuda().contiguous()
    layer_mode = "layer" in showerMap
    le = None
    if layer_mode:
        assert layerE is not None
        le = torch.as_tensor(np.asarray(layerE, dtype=np.float32)).cuda().contiguous()
        data = data.squeeze()
        if data.dim() != 3 or le.shape != (B, data.shape[1] + 1):
            raise ValueError("ReverseNormHGCal: the layer renormalisation works on decoded showers (batch, layers, cells) with "
                             "layerE (batch, 1 + layers); got %s and %s" % (tuple(data.shape), tuple(le.shape)))
        dims = (data.shape[1], 1, data.shape[2])
    else:
        dims = (1, 1, int(np.prod(data.shape[1:])))
    en = torch.as_tensor(np.ascontiguousarray(energy.reshape(B), dtype=np.float32)).cuda()
    out = _staged(data.contiguous(), en, le, dims, c, max_deposit, 1e-8, 1e-8, 2)
    return out.reshape(data.shape).cpu().numpy(), gen_out


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
