"""ctypes binding of ``lib/libcalodiff_hip.so`` (C ABI: ``include/calodiff.h``).

PyTorch is used here for device memory, streams and parameter storage only.  There is no fallback of
any kind: if the library is missing, or no gfx950 GPU is visible, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CALODIFF_LIB") or os.path.join(_HERE, "lib", "libcalodiff_hip.so")  # override: A/B builds

CD_MAX_SIZES = 8
CD_ABI_VERSION = 3  # include/calodiff.h; load_library refuses a library that reports another one
TIME_KINDS = {"log": 0, "sigma": 1, "raw": 2}
OBJECTIVES = {"hybrid": 0, "noise_pred": 1, "mean_pred": 2}
LOSS_TYPES = {"l2": 0, "l1": 1, "mse": 2, "huber": 3}  # CD_LOSS_* (Loss._loss, models/loss.py:97-116)


class CdUnetDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("grid", C.c_int32 * 3),
        ("in_channels", C.c_int32),
        ("n_sizes", C.c_int32),
        ("layer_sizes", C.c_int32 * CD_MAX_SIZES),
        ("groups", C.c_int32),
        ("block_attn", C.c_int32),
        ("mid_attn", C.c_int32),
        ("compress_z", C.c_int32),
        ("cond_size", C.c_int32),
        ("cond_dim", C.c_int32),
        ("rz_input", C.c_int32),
        ("phi_input", C.c_int32),
        ("time_embed_kind", C.c_int32),
        ("objective", C.c_int32),
        ("sigma_data", C.c_float),
        ("time_sin", C.c_int32),
        ("cond_sin", C.c_int32),
    ]


SOP_LINCOMB, SOP_DENOISE, SOP_RANDN, SOP_RECORD, SOP_LINDIV = 0, 1, 2, 3, 4


class CdSamplerOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dst", C.c_int32), ("nsrc", C.c_int32), ("src", C.c_int32 * 6), ("col", C.c_int32)]


class CdLayerMlpDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dim_in", C.c_int32), ("hidden", C.c_int32), ("cond_emb", C.c_int32), ("cond_size", C.c_int32),
                ("n_res", C.c_int32), ("time_embed_kind", C.c_int32), ("objective", C.c_int32), ("sigma_data", C.c_float)]


class CdStep(C.Structure):
    _fields_ = [("sigma", C.c_float), ("sigma_prev_masked", C.c_float), ("ddim_sigma", C.c_float), ("denom", C.c_float)]


# every symbol include/calodiff.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGNATURES = {
    "cd_last_error": (C.c_char_p, []),
    "cd_abi_version": (C.c_int, []),
    "cd_device_check": (C.c_int, [C.c_char_p, C.c_int]),
    "cd_plan_create": (C.c_int, [C.POINTER(CdUnetDesc), C.POINTER(_P)]),
    "cd_plan_destroy": (C.c_int, [_P]),
    "cd_plan_num_weights": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "cd_plan_weight_name": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "cd_plan_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "cd_plan_set_weights": (C.c_int, [_P, C.c_int, C.POINTER(_P), _P]),
    "cd_plan_set_coords": (C.c_int, [_P, _P, _P, _P, _P]),
    "cd_plan_workspace_bytes": (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    "cd_unet_forward": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cd_denoise": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cd_denoise_safe": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, C.POINTER(C.c_int), _P]),
    "cd_ddim_sample": (C.c_int, [_P, C.c_int, _P, _P, C.POINTER(CdStep), C.c_int, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P,
                                 C.c_int, _P, C.c_size_t, _P]),
    "cd_plan_sampler_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "cd_sampler_run": (C.c_int, [_P, C.c_int, _P, C.c_float, _P, C.c_int, C.c_int, C.POINTER(CdSamplerOp), C.c_int,
                                 C.POINTER(C.c_int32), _P, C.c_int, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, C.c_int,
                                 _P, C.c_size_t, _P]),
    "cd_randn": (C.c_int, [_P, C.c_int64, C.c_uint64, C.c_uint64, _P]),
    "cd_plan_grad_layout": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "cd_plan_train_workspace_bytes": (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    "cd_plan_status": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "cd_layer_forward": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "cd_layer_denoise": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "cd_layer_sample": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "cd_layer_train_workspace_bytes": (C.c_int, [_P, C.c_int, C.POINTER(C.c_size_t)]),
    "cd_layer_train_step": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cd_layer_train_step_loss": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, C.c_size_t,
                                           _P]),
    "cd_reverse_norm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_float, C.c_float, _P]),
    "cd_reverse_norm_staged": (C.c_int, [_P, _P, _P, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int, _P]),
    "cd_adam_step": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int64), C.c_double, C.c_double, C.c_double, C.c_float, C.c_float, C.c_int, _P]),
    "cd_train_step": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "cd_set_conv_precision": (C.c_int, [C.c_char_p]),
    "cd_get_conv_precision": (C.c_char_p, []),
    "cd_profile_begin": (C.c_int, []),
    "cd_profile_end": (C.c_int, [C.c_char_p, C.c_int]),
    "cd_loss_hybrid_l2": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cd_loss_hybrid": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "cd_op_to_channels_last": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, _P]),
    "cd_op_to_ncdhw": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, _P]),
    "cd_op_cyl_conv": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, _P]),
    "cd_op_cyl_conv_transpose": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                           C.POINTER(C.c_int32), _P, _P]),
    "cd_op_init_conv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), _P, _P]),
    "cd_op_group_norm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, _P, _P]),
    "cd_op_resnet_block": (C.c_int, [_P, C.c_int, _P, C.c_int, C.POINTER(_P), _P, _P, C.c_int, C.c_int,
                                     C.POINTER(C.c_int32), C.c_int, _P, C.c_size_t, _P]),
    "cd_op_linear_attention": (C.c_int, [_P, C.POINTER(_P), _P, C.c_int, C.c_int, C.POINTER(C.c_int32), _P, C.c_size_t, _P]),
    "cd_op_conv_backward": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, C.c_size_t, _P]),
    "cd_op_conv_transpose_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                                C.POINTER(C.c_int32), _P, C.c_size_t, _P]),
    "cd_op_group_norm_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _P,
                                            C.c_size_t, _P]),
    "cd_op_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int64]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load_library() -> C.CDLL:
    """dlopen the HIP library and bind every declared symbol; fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m calodiffusion_amd.build` (hipcc, --offload-arch=gfx950). "
            "calodiffusion_amd has no CPU or PyTorch fallback.")
    _check_built_from_these_sources()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    if lib.cd_abi_version() != CD_ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI version {lib.cd_abi_version()}, this binding is written against "
                           f"{CD_ABI_VERSION} (include/calodiff.h): rebuild with `python -m calodiffusion_amd.build`")
    _lib = lib
    return lib


def _check_built_from_these_sources():
    """The library is git-ignored and travels between boxes as a built file: refuse one that was not built from the sources
    next to it (a stale .so with an older argument list shifts every pointer argument).  build.py writes `<lib>.srchash`
    after every link; CALODIFF_LIB (an explicit A/B build) and CD_SKIP_SRCHASH=1 bypass the check."""
    if os.environ.get("CALODIFF_LIB") or os.environ.get("CD_SKIP_SRCHASH"):
        return
    from . import build
    stamp = LIB_PATH + ".srchash"
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    want = build.source_hash()
    if have != want:
        raise RuntimeError(f"{LIB_PATH} was not built from the sources in {build.CSRC} (stamp {str(have)[:12]}, sources "
                           f"{want[:12]}): run `python -m calodiffusion_amd.build`")


def _check(code: int):
    if code != 0:
        msg = load_library().cd_last_error().decode(errors="replace")
        exc = ValueError if code == -1 else RuntimeError
        raise exc(f"calodiff[{code}]: {msg}")


def require_gpu() -> str:
    lib = load_library()
    if not torch.cuda.is_available():
        raise RuntimeError("calodiffusion_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and "
                           "there is no CPU fallback")
    buf = C.create_string_buffer(256)
    _check(lib.cd_device_check(buf, 256))
    return buf.value.decode()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(ROCm) tensor: calodiffusion_amd computes on the GPU only")
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


def _i32x3(v: Sequence[int]):
    return (C.c_int32 * 3)(*[int(a) for a in v])


class UnetEngine:
    """One HIP plan bound to one CondUnet parameter set on one device."""

    def __init__(self, unet, rz_input=False, phi_input=False, time_kind="raw", objective="hybrid", sigma_data=1.0,
                 coords=None):
        self.lib = load_library()
        self.device_name = require_gpu()
        self.unet = unet
        d = CdUnetDesc()
        d.struct_size = C.sizeof(CdUnetDesc)
        d.grid = _i32x3(unet.grid)
        d.in_channels = unet.channels
        d.n_sizes = len(unet.layer_sizes)
        for i, v in enumerate(unet.layer_sizes):
            d.layer_sizes[i] = int(v)
        d.groups = unet.groups
        d.block_attn = int(bool(unet.block_attn))
        d.mid_attn = int(unet.mid_attn is not False)
        d.compress_z = int(bool(unet.compress_Z))
        d.cond_size, d.cond_dim = unet.cond_size, unet.cond_dim
        d.rz_input, d.phi_input = int(bool(rz_input)), int(bool(phi_input))
        d.time_embed_kind = TIME_KINDS[time_kind]
        d.objective = OBJECTIVES[objective]
        d.sigma_data = float(sigma_data)
        d.time_sin, d.cond_sin = int(bool(getattr(unet, "time_embed", False))), int(bool(getattr(unet, "cond_embed", False)))
        self.desc = d
        self.grid = tuple(unet.grid)
        self.voxels = int(np.prod(self.grid))
        handle = _P()
        _check(self.lib.cd_plan_create(C.byref(d), C.byref(handle)))
        self.plan = handle
        self._weights_version = None
        self._weight_order = None  # [(state_dict name, tensor)] in the plan's order
        self._weight_ids = None
        self._held_weights = []
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.device = next(unet.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("move the model to the GPU first (model.to('cuda')): calodiffusion_amd has no CPU path")
        if coords is not None:
            r, z, phi = (np.ascontiguousarray(a, dtype=np.float32) for a in coords)
            assert len(r) == self.grid[2] and len(z) == self.grid[0] and len(phi) == self.grid[1]
            _check(self.lib.cd_plan_set_coords(self.plan, r.ctypes.data, z.ctypes.data, phi.ctypes.data, _stream()))
        self.sync_weights(force=True)

    @classmethod
    def for_unet(cls, unet):
        opts = getattr(unet, "_engine_opts", {})
        return cls(unet, **opts)

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.cd_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _version(self):
        ps = list(self.unet.parameters())
        return tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps) + (tuple(id(p) for p in ps),)

    def sync_weights(self, force=False):
        """(Re-)pack the parameters into the plan's arena; cheap no-op when nothing changed.  All tensors go in ONE C-ABI call
        (cd_plan_set_weights: two launches); the plan's tensor order and the Parameter objects behind it are looked up once (and
        again whenever a Parameter object of the module is replaced)."""
        ver = self._version()
        if not force and ver == self._weights_version:
            return
        ids = ver[-1]
        if self._weight_order is None or self._weight_ids != ids:
            sd = self.unet.state_dict(keep_vars=True)
            n = C.c_int()
            _check(self.lib.cd_plan_num_weights(self.plan, C.byref(n)))
            buf = C.create_string_buffer(256)
            numel = C.c_int64()
            order = []
            for i in range(n.value):
                _check(self.lib.cd_plan_weight_name(self.plan, i, buf, 256, C.byref(numel)))
                name = buf.value.decode()
                if name not in sd:
                    raise RuntimeError(f"state_dict has no tensor named {name!r}")
                if sd[name].numel() != numel.value:
                    raise RuntimeError(f"{name}: {sd[name].numel()} elements, HIP plan expects {numel.value}")
                order.append((name, sd[name]))
            missing = set(sd) - {name for name, _ in order}
            if missing:
                raise RuntimeError(f"HIP plan does not consume these state_dict tensors: {sorted(missing)[:5]} ...")
            self._weight_order, self._weight_ids = order, ids
        tensors = [_dev32(t.detach(), name) for name, t in self._weight_order]
        ptrs = (_P * len(tensors))(*[t.data_ptr() for t in tensors])
        _check(self.lib.cd_plan_set_weights(self.plan, len(tensors), ptrs, _stream()))
        self._held_weights = tensors  # (conversions made by _dev32 must outlive the launches)
        self._weights_version = ver

    # launch-sequence switches the library reads per call (tests and A/B runs flip them inside one process): a workspace sized
    # under one setting is not valid under another, so they are part of the cache key.  (The arithmetic mode is not: the
    # library sizes for the largest of its three modes.)
    _WS_SWITCHES = ("CD_NO_DEEP_LEVEL", "CD_NO_PW_CLOSE", "CD_NO_FUSED_ATTN", "CD_NO_GNDEFER", "CD_ATTN_COMBINE_LAUNCH", "CD_PW_F32",
                    "CD_NO_ATTN_MOMENTS", "CD_ATTN_MOM_MIN")

    def _ws_key(self, batch: int):
        return (int(batch),) + tuple(os.environ.get(k) for k in self._WS_SWITCHES)

    def workspace(self, batch: int) -> torch.Tensor:
        key = ("net",) + self._ws_key(batch)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = C.c_size_t()
            _check(self.lib.cd_plan_workspace_bytes(self.plan, batch, C.byref(nbytes)))
            # keep ONE network workspace and ONE sampler-program workspace (each ~1 GB at the headline batch: alternating
            # denoise / sampler calls would otherwise reallocate every time, a ragged last batch would double the footprint)
            self._ws = {k: v for k, v in self._ws.items() if k[0] != "net"}
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ compute
    def unet_forward(self, x, cond, time):
        x, cond, time = _dev32(x, "x"), _dev32(cond, "cond"), _dev32(time, "time")
        B = x.shape[0]
        if tuple(x.shape[1:]) != (self.unet.channels,) + self.grid:
            raise ValueError(f"x has shape {tuple(x.shape)}, expected (B, {self.unet.channels}, {self.grid})")
        if self.desc.cond_sin:  # sinusoidal cond embedding: one scalar per sample (models.py:132-144 broadcasts a (B,) tensor)
            if cond.numel() != B:
                raise ValueError("cond_embed='sin' takes cond of shape (B,)")
        elif cond.shape != (B, self.unet.cond_size):
            raise ValueError("cond must be (B, cond_size)")
        if time.numel() != B:
            raise ValueError("time must be (B,)")
        self.sync_weights()
        ws = self.workspace(B)
        out = torch.empty((B, 1) + self.grid, dtype=torch.float32, device=x.device)
        _check(self.lib.cd_unet_forward(self.plan, B, x.data_ptr(), cond.data_ptr(), time.data_ptr(), out.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream()))
        return out

    # denoise() recovers from an fp16-range overflow by itself (cd_denoise_safe: one stream synchronisation per call to read the
    # flag, then a full-range re-run if it was raised) so that samplers calling the model back from Python never die
    # mid-trajectory; set False for the asynchronous, graph-capturable cd_denoise, whose overflow only shows in check_status().
    safe_denoise = True

    def denoise(self, x, sigma, cond):
        x, cond = _dev32(x, "x"), _dev32(cond, "cond")
        B = x.shape[0]
        sigma = _dev32(sigma, "sigma").reshape(-1)
        if sigma.numel() == 1 and B > 1:
            sigma = sigma.expand(B).contiguous()
        if tuple(x.shape[1:]) != (1,) + self.grid or sigma.numel() != B or cond.shape != (B, self.unet.cond_size):
            raise ValueError(f"denoise shapes: x {tuple(x.shape)}, sigma {tuple(sigma.shape)}, cond {tuple(cond.shape)}")
        self.sync_weights()
        ws = self.workspace(B)
        out = torch.empty_like(x)
        if self.safe_denoise:
            fell = C.c_int(0)
            _check(self.lib.cd_denoise_safe(self.plan, B, x.data_ptr(), sigma.data_ptr(), cond.data_ptr(), out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), C.byref(fell), _stream()))
            if fell.value:
                self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
        else:
            _check(self.lib.cd_denoise(self.plan, B, x.data_ptr(), sigma.data_ptr(), cond.data_ptr(), out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _stream()))
        return out

    def ddim_sample(self, start, cond, steps: np.ndarray, step_noise=None, seed=0, offset=0, debug=False, use_graph=True,
                    out=None, noise_stride=0):
        """steps: float32 array (n_steps, 4) = (sigma, sigma_prev*[t>0], ddim_sigma, denom).  noise_stride: Philox stream
        distance between the noise tensors of consecutive steps (0 = this batch's own size; a batch shard passes the global
        tensor size so that the union of the shards is the single-GPU result)."""
        start, cond = _dev32(start, "start"), _dev32(cond, "cond")
        B = start.shape[0]
        steps = np.ascontiguousarray(steps, dtype=np.float32)
        n_steps = steps.shape[0]
        assert steps.shape == (n_steps, 4)
        self.sync_weights()
        ws = self.workspace(B)
        x_out = torch.empty_like(start) if out is None else out
        xs = x0s = None
        if debug:
            xs = torch.empty((n_steps,) + tuple(start.shape), dtype=torch.float32, device=start.device)
            x0s = torch.empty_like(xs)
        if step_noise is not None:
            step_noise = _dev32(step_noise, "step_noise")
            assert step_noise.shape[0] == n_steps and step_noise[0].numel() == start.numel()
        _check(self.lib.cd_ddim_sample(self.plan, B, start.data_ptr(), cond.data_ptr(),
                                       steps.ctypes.data_as(C.POINTER(CdStep)), n_steps, _ptr(step_noise), int(seed),
                                       int(offset), int(noise_stride), x_out.data_ptr(), _ptr(xs), _ptr(x0s),
                                       int(bool(use_graph)), ws.data_ptr(), ws.numel(), _stream()))
        self.check_status()
        return x_out, xs, x0s

    def sampler_run(self, start, cond, program, step_noise=None, seed=0, offset=0, noise_stride=0, debug=False, use_graph=True):
        """Run a sampler step program (calodiffusion_amd.sample.Program) on the device loop (cd_sampler_run).
        Returns (x, xs, x0s); the trajectories (n_steps, B, 1, D, H, W) only when ``debug`` and the program records them."""
        start, cond = _dev32(start, "start"), _dev32(cond, "cond")
        B = start.shape[0]
        coefs = np.ascontiguousarray(program.coefs, dtype=np.float32)
        n_steps, n_coef = coefs.shape
        ops = (CdSamplerOp * len(program.ops))()
        for o, (kind, dst, src, col) in zip(ops, program.ops):
            o.kind, o.dst, o.nsrc, o.col = kind, dst, len(src), col
            for j, v in enumerate(src):
                o.src[j] = v
        op_begin = None
        if program.op_begin is not None:
            assert len(program.op_begin) == n_steps + 1
            op_begin = (C.c_int32 * (n_steps + 1))(*program.op_begin)
        self.sync_weights()
        nbytes = C.c_size_t()
        _check(self.lib.cd_plan_sampler_workspace_bytes(self.plan, B, program.n_bufs, n_steps, n_coef, C.byref(nbytes)))
        key = ("sampler",) + self._ws_key(B)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes.value:
            self._ws = {k: v for k, v in self._ws.items() if k[0] != "sampler"}  # (other batch sizes / switch settings: evicted)
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        x_out = torch.empty_like(start)
        xs = x0s = None
        if debug:
            kinds = [(k, d) for k, d, _, _ in program.ops if k == SOP_RECORD]
            if (SOP_RECORD, 0) in kinds:
                xs = torch.zeros((n_steps,) + tuple(start.shape), dtype=torch.float32, device=start.device)
            if (SOP_RECORD, 1) in kinds:
                x0s = torch.zeros((n_steps,) + tuple(start.shape), dtype=torch.float32, device=start.device)
        if step_noise is not None:
            step_noise = _dev32(step_noise, "step_noise")
            assert step_noise.numel() == program.n_randn * start.numel(), "step_noise: one (B,1,D,H,W) tensor per RANDN op executed"
        _check(self.lib.cd_sampler_run(self.plan, B, start.data_ptr(), float(program.start_scale), cond.data_ptr(), program.n_bufs,
                                       n_steps, ops, len(program.ops), op_begin, coefs.ctypes.data, n_coef, _ptr(step_noise),
                                       int(seed), int(offset), int(noise_stride), x_out.data_ptr(), _ptr(xs), _ptr(x0s),
                                       int(bool(use_graph)), ws.data_ptr(), ws.numel(), _stream()))
        self.check_status()
        return x_out, xs, x0s

    def check_status(self):
        """Raise if a compute call since the last check left the fp16 range of the f16x2 convolution path (synchronises).
        The sampler entry points recover by themselves (bf16x3 re-run of the trajectory): that only sets ``range_fallbacks``."""
        flags = C.c_int(0)
        _check(self.lib.cd_plan_status(self.plan, C.byref(flags), _stream()))
        if flags.value & 2:
            self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
        if flags.value & 1:
            raise FloatingPointError("calodiff: an activation exceeded the fp16 range of the f16x2 convolution kernels "
                                     "(outputs contain inf/NaN); set CD_CONV_PRECISION=bf16x3 for the full fp32 range")

    # ------------------------------------------------------------------ training
    def grad_layout(self):
        """{state_dict name: (offset, numel)} into the flat gradient buffer, and its total length."""
        if getattr(self, "_grad_layout", None) is None:
            n = C.c_int()
            _check(self.lib.cd_plan_num_weights(self.plan, C.byref(n)))
            buf = C.create_string_buffer(256)
            numel, off, total = C.c_int64(), C.c_int64(), C.c_int64()
            lay = {}
            for i in range(n.value):
                _check(self.lib.cd_plan_weight_name(self.plan, i, buf, 256, C.byref(numel)))
                _check(self.lib.cd_plan_grad_layout(self.plan, i, C.byref(off), C.byref(total)))
                lay[buf.value.decode()] = (off.value, numel.value)
            self._grad_layout = (lay, total.value)
        return self._grad_layout

    def train_workspace(self, batch: int) -> torch.Tensor:
        ws = getattr(self, "_tws", {}).get(batch)
        if ws is None:
            nbytes = C.c_size_t()
            _check(self.lib.cd_plan_train_workspace_bytes(self.plan, batch, C.byref(nbytes)))
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
            self._tws = {batch: ws}
        return ws

    def train_step(self, data, noise, sigma, cond, loss_type="l2"):
        """hybrid_weight loss (LOSS_TYPE l2 / l1 / mse / huber) and the gradient of every parameter (flat fp32 buffer, see
        grad_layout)."""
        data, noise, cond = _dev32(data, "data"), _dev32(noise, "noise"), _dev32(cond, "cond")
        sigma = _dev32(sigma, "sigma").reshape(-1)
        B = data.shape[0]
        self.sync_weights()
        ws = self.train_workspace(B)
        _, total = self.grad_layout()
        flat = torch.empty(total, dtype=torch.float32, device=data.device)
        loss = torch.empty((), dtype=torch.float64, device=data.device)
        _check(self.lib.cd_train_step(self.plan, B, data.data_ptr(), noise.data_ptr(), sigma.data_ptr(), cond.data_ptr(),
                                      LOSS_TYPES[loss_type], loss.data_ptr(), flat.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        return loss, flat

    def param_grads(self, flat):
        """Views of the flat gradient buffer, one per parameter of the bound CondUnet, in .parameters() order."""
        lay, _ = self.grad_layout()
        out = []
        for name, p in self.unet.named_parameters():
            off, numel = lay[name]
            out.append(flat[off:off + numel].view(p.shape))
        return out

    def loss_hybrid(self, data, noise, sigma, cond, loss_type="l2"):
        data, noise, cond = _dev32(data, "data"), _dev32(noise, "noise"), _dev32(cond, "cond")
        sigma = _dev32(sigma, "sigma").reshape(-1)
        B = data.shape[0]
        self.sync_weights()
        ws = self.workspace(B)
        out = torch.empty((), dtype=torch.float64, device=data.device)
        _check(self.lib.cd_loss_hybrid(self.plan, B, data.data_ptr(), noise.data_ptr(), sigma.data_ptr(), cond.data_ptr(),
                                       LOSS_TYPES[loss_type], out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
        return out

    def loss_hybrid_l2(self, data, noise, sigma, cond):
        return self.loss_hybrid(data, noise, sigma, cond, "l2")


class LayerMlpEngine:
    """The layer-energy MLP of LayerDiffusion on the HIP library (cd_layer_forward / cd_layer_denoise / cd_layer_sample).
    Stateless on the library side: the parameters are read in place from the torch storage."""

    def __init__(self, resnet, time_kind="raw", objective="hybrid", sigma_data=1.0):
        self.lib = load_library()
        self.device_name = require_gpu()
        self.net = resnet
        d = CdLayerMlpDesc()
        d.struct_size = C.sizeof(CdLayerMlpDesc)
        d.dim_in, d.hidden, d.cond_emb, d.cond_size = resnet.dim_in, resnet.hidden_dim, resnet.cond_emb_dim, resnet.cond_size
        d.n_res = len(resnet.hidden_layers)
        d.time_embed_kind, d.objective, d.sigma_data = TIME_KINDS[time_kind], OBJECTIVES[objective], float(sigma_data)
        self.desc = d
        self.dim = resnet.dim_in

    def _weights(self):
        ps = [_dev32(p.detach(), k) for k, p in self.net.state_dict().items()]
        if len(ps) != 2 * (8 + 3 * self.desc.n_res):
            raise RuntimeError("unexpected layer-model state_dict")
        for p in ps:
            if p.data_ptr() % 16:
                raise RuntimeError("layer-model parameters must be 16-byte aligned")
        self._keep = ps
        return (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps]), len(ps)

    def _io(self, x, cond):
        x, cond = _dev32(x, "x"), _dev32(cond, "cond")
        B = x.shape[0]
        if x.shape != (B, self.dim) or cond.shape != (B, self.desc.cond_size):
            raise ValueError(f"layer model shapes: x {tuple(x.shape)} (expected (B, {self.dim})), cond {tuple(cond.shape)} "
                             f"(expected (B, {self.desc.cond_size}))")
        return x, cond, B

    def forward(self, x, cond, time):
        x, cond, B = self._io(x, cond)
        time = _dev32(time, "time").reshape(-1)
        if time.numel() != B:
            raise ValueError("time must be (B,)")
        w, n = self._weights()
        out = torch.empty_like(x)
        _check(self.lib.cd_layer_forward(C.byref(self.desc), w, n, B, x.data_ptr(), cond.data_ptr(), time.data_ptr(),
                                         out.data_ptr(), _stream()))
        return out

    def denoise(self, x, sigma, cond):
        x, cond, B = self._io(x, cond)
        sigma = _dev32(sigma, "sigma").reshape(-1)
        if sigma.numel() == 1 and B > 1:
            sigma = sigma.expand(B).contiguous()
        if sigma.numel() != B:
            raise ValueError("sigma must be (B,)")
        w, n = self._weights()
        out = torch.empty_like(x)
        _check(self.lib.cd_layer_denoise(C.byref(self.desc), w, n, B, x.data_ptr(), sigma.data_ptr(), cond.data_ptr(),
                                         out.data_ptr(), _stream()))
        return out

    # ------------------------------------------------------------------ training (same contract as UnetEngine's)
    def grad_layout(self):
        """{state_dict name: (offset, numel)} into the flat gradient buffer (state_dict order), and its total length."""
        lay, off = {}, 0
        for k, p in self.net.state_dict().items():
            lay[k] = (off, p.numel())
            off += p.numel()
        return lay, off

    def train_step(self, data, noise, sigma, cond, loss_type="l2"):
        """hybrid_weight loss (LOSS_TYPE l2 / l1 / mse / huber) and the gradient of every parameter of the layer model
        (cd_layer_train_step_loss)."""
        data, cond, B = self._io(data, cond)
        noise = _dev32(noise, "noise")
        sigma = _dev32(sigma, "sigma").reshape(-1)
        if noise.shape != data.shape or sigma.numel() != B:
            raise ValueError("noise must have the shape of data and sigma must be (B,)")
        w, n = self._weights()
        nbytes = C.c_size_t()
        _check(self.lib.cd_layer_train_workspace_bytes(C.byref(self.desc), B, C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=data.device)
        _, total = self.grad_layout()
        flat = torch.empty(total, dtype=torch.float32, device=data.device)
        loss = torch.empty((), dtype=torch.float64, device=data.device)
        _check(self.lib.cd_layer_train_step_loss(C.byref(self.desc), w, n, B, data.data_ptr(), noise.data_ptr(), sigma.data_ptr(),
                                                 cond.data_ptr(), LOSS_TYPES[loss_type], loss.data_ptr(), flat.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _stream()))
        return loss, flat

    def param_grads(self, flat):
        """Views of the flat gradient buffer, one per parameter of the bound ResNet, in .parameters() order."""
        lay, _ = self.grad_layout()
        names = {id(p): k for k, p in self.net.named_parameters()}
        return [flat[lay[names[id(p)]][0]: lay[names[id(p)]][0] + p.numel()].view_as(p) for p in self.net.parameters()]

    def loss_hybrid(self, data, noise, sigma, cond, loss_type="l2"):
        return self.train_step(data, noise, sigma, cond, loss_type)[0].to(torch.float32)

    def loss_hybrid_l2(self, data, noise, sigma, cond):
        return self.loss_hybrid(data, noise, sigma, cond, "l2")

    def ddim_sample(self, start, cond, steps: np.ndarray, step_noise=None, seed=0, offset=0, debug=False, use_graph=True,
                    out=None, noise_stride=0):
        """Same contract as UnetEngine.ddim_sample; the whole trajectory is one launch (use_graph is irrelevant; the noise of
        a stochastic sampler is drawn as one (n_steps, B, dim) block, so batch shards are not slices of a global stream)."""
        start, cond, B = self._io(start, cond)
        steps = np.ascontiguousarray(steps, dtype=np.float32)
        n_steps = steps.shape[0]
        assert steps.shape == (n_steps, 4)
        table = torch.from_numpy(steps).to(start.device)
        if step_noise is None and float(np.abs(steps[:, 2]).max()) > 0:
            step_noise = randn((n_steps, B, self.dim), start.device, seed, offset)
        if step_noise is not None:
            step_noise = _dev32(step_noise, "step_noise")
            assert step_noise.numel() == n_steps * start.numel()
        x_out = torch.empty_like(start) if out is None else out
        xs = x0s = None
        if debug:
            xs = torch.empty((n_steps,) + tuple(start.shape), dtype=torch.float32, device=start.device)
            x0s = torch.empty_like(xs)
        w, n = self._weights()
        _check(self.lib.cd_layer_sample(C.byref(self.desc), w, n, B, start.data_ptr(), cond.data_ptr(), table.data_ptr(),
                                        n_steps, _ptr(step_noise), x_out.data_ptr(), _ptr(xs), _ptr(x0s), _stream()))
        return x_out, xs, x0s


def set_conv_precision(mode: str):
    """'f16x2' (default), 'bf16x3' or 'f32' arithmetic of the convolutions, process-wide (cd_set_conv_precision)."""
    _check(load_library().cd_set_conv_precision(mode.encode()))


def get_conv_precision() -> str:
    return load_library().cd_get_conv_precision().decode()


def profile_begin():
    _check(load_library().cd_profile_begin())


def profile_end() -> dict:
    import json
    buf = C.create_string_buffer(1 << 18)
    _check(load_library().cd_profile_end(buf, 1 << 18))
    return json.loads(buf.value.decode())


def randn(shape, device, seed: int, offset: int = 0) -> torch.Tensor:
    """Unit normals from the device Philox stream (element i is a function of (seed, offset + i) only)."""
    lib = load_library()
    require_gpu()
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    _check(lib.cd_randn(out.data_ptr(), out.numel(), int(seed), int(offset), _stream()))
    return out


# ---------------------------------------------------------------------- primitive ops (parity tests)
class Ops:
    """Thin wrappers over the cd_op_* entry points.  Activations are channels-last (B, D, H, W, C)."""

    def __init__(self):
        self.lib = load_library()
        require_gpu()
        self._scratch = None

    def scratch(self, batch, channels, voxels):
        need = self.lib.cd_op_scratch_bytes(batch, channels, voxels)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device="cuda")
        return self._scratch

    def to_channels_last(self, x):
        x = _dev32(x, "x")
        B, Cc = x.shape[:2]
        vox = int(np.prod(x.shape[2:]))
        y = torch.empty((B,) + tuple(x.shape[2:]) + (Cc,), dtype=torch.float32, device=x.device)
        _check(self.lib.cd_op_to_channels_last(x.data_ptr(), y.data_ptr(), B, Cc, vox, _stream()))
        return y

    def to_ncdhw(self, y):
        y = _dev32(y, "y")
        B, Cc = y.shape[0], y.shape[-1]
        vox = int(np.prod(y.shape[1:-1]))
        x = torch.empty((B, Cc) + tuple(y.shape[1:-1]), dtype=torch.float32, device=y.device)
        _check(self.lib.cd_op_to_ncdhw(y.data_ptr(), x.data_ptr(), B, Cc, vox, _stream()))
        return x

    def cyl_conv(self, x_cl, w, bias, stride=(1, 1, 1), x1_cl=None):
        x_cl, w = _dev32(x_cl, "x"), _dev32(w, "w")
        B, D, H, W, c0 = x_cl.shape
        c1 = 0 if x1_cl is None else x1_cl.shape[-1]
        cout = w.shape[0]
        k = tuple(w.shape[2:])
        if k == (1, 1, 1):
            od = (D, H, W)
        else:
            od = ((D + 2 - k[0]) // stride[0] + 1, (H + 2 - k[1]) // stride[1] + 1, (W + 2 - k[2]) // stride[2] + 1)
        y = torch.empty((B,) + od + (cout,), dtype=torch.float32, device=x_cl.device)
        sc = self.scratch(B, max(c0 + c1, cout), D * H * W)
        _check(self.lib.cd_op_cyl_conv(x_cl.data_ptr(), c0, _ptr(x1_cl), c1, w.data_ptr(), _ptr(bias), y.data_ptr(), B, cout,
                                       _i32x3((D, H, W)), _i32x3(k), _i32x3(stride), sc.data_ptr(), _stream()))
        return y

    def cyl_conv_transpose(self, x_cl, w, bias, kernel_z, stride_z, out_pad):
        x_cl, w = _dev32(x_cl, "x"), _dev32(w, "w")
        B, D, H, W, c = x_cl.shape
        od = ((D - 1) * stride_z - 2 + kernel_z, 2 * H + out_pad[1], 2 * W + out_pad[2])
        y = torch.empty((B,) + od + (c,), dtype=torch.float32, device=x_cl.device)
        sc = self.scratch(B, c, int(np.prod(od)))
        _check(self.lib.cd_op_cyl_conv_transpose(x_cl.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, c,
                                                 _i32x3((D, H, W)), kernel_z, stride_z, _i32x3(out_pad), sc.data_ptr(),
                                                 _stream()))
        return y

    def init_conv(self, x_ncdhw, w, bias):
        x, w = _dev32(x_ncdhw, "x"), _dev32(w, "w")
        B, cin, D, H, W = x.shape
        cout = w.shape[0]
        y = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=x.device)
        sc = self.scratch(B, cout, D * H * W)
        _check(self.lib.cd_op_init_conv(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), B, cin, cout, _i32x3((D, H, W)),
                                        sc.data_ptr(), _stream()))
        return y

    def group_norm(self, x_cl, gamma, beta, groups, silu=False, add_bc=None, residual=None):
        x = _dev32(x_cl, "x")
        B, Cc = x.shape[0], x.shape[-1]
        vox = int(np.prod(x.shape[1:-1]))
        y = torch.empty_like(x)
        sc = self.scratch(B, Cc, vox)
        _check(self.lib.cd_op_group_norm(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), B, Cc, vox, groups,
                                         int(silu), _ptr(add_bc), _ptr(residual), sc.data_ptr(), _stream()))
        return y

    def _bws(self, nbytes=1 << 30):
        if getattr(self, "_bwd_ws", None) is None or self._bwd_ws.numel() < nbytes:
            self._bwd_ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        return self._bwd_ws

    def conv_backward(self, x_cl, w, dy_cl, stride=(1, 1, 1), x1_cl=None, need_dx=True, bias=True):
        """Gradients (dx, dw, db) of cyl_conv; channels-last activations, torch-layout weights."""
        x, w, dy = _dev32(x_cl, "x"), _dev32(w, "w"), _dev32(dy_cl, "dy")
        B, D, H, W, c0 = x.shape
        c1 = 0 if x1_cl is None else x1_cl.shape[-1]
        cout, k = w.shape[0], tuple(w.shape[2:])
        dx = torch.empty((B, D, H, W, c0 + c1), dtype=torch.float32, device=x.device) if need_dx else None
        dw = torch.empty_like(w)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if bias else None
        ws = self._bws()
        _check(self.lib.cd_op_conv_backward(x.data_ptr(), c0, _ptr(x1_cl), c1, w.data_ptr(), dy.data_ptr(), _ptr(dx), dw.data_ptr(),
                                            _ptr(db), B, cout, _i32x3((D, H, W)), _i32x3(k), _i32x3(stride), ws.data_ptr(),
                                            ws.numel(), _stream()))
        return dx, dw, db

    def conv_transpose_backward(self, x_cl, w, dy_cl, kernel_z, stride_z, out_pad):
        x, w, dy = _dev32(x_cl, "x"), _dev32(w, "w"), _dev32(dy_cl, "dy")
        B, D, H, W, c = x.shape
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty((c,), dtype=torch.float32, device=x.device)
        ws = self._bws()
        _check(self.lib.cd_op_conv_transpose_backward(x.data_ptr(), w.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                      db.data_ptr(), B, c, _i32x3((D, H, W)), kernel_z, stride_z, _i32x3(out_pad),
                                                      ws.data_ptr(), ws.numel(), _stream()))
        return dx, dw, db

    def group_norm_backward(self, x_cl, gamma, beta, dy_cl, groups, silu=False, want_dadd=False):
        x, dy = _dev32(x_cl, "x"), _dev32(dy_cl, "dy")
        B, Cc = x.shape[0], x.shape[-1]
        vox = int(np.prod(x.shape[1:-1]))
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        dadd = torch.empty((B, Cc), dtype=torch.float32, device=x.device) if want_dadd else None
        ws = self._bws()
        _check(self.lib.cd_op_group_norm_backward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                                  dg.data_ptr(), db.data_ptr(), _ptr(dadd), B, Cc, vox, groups, int(silu),
                                                  ws.data_ptr(), ws.numel(), _stream()))
        return dx, dg, db, dadd

    _RES_KEYS = ("block1.proj.conv.weight", "block1.proj.conv.bias", "block1.norm.weight", "block1.norm.bias",
                 "block2.proj.conv.weight", "block2.proj.conv.bias", "block2.norm.weight", "block2.norm.bias",
                 "mlp.1.weight", "mlp.1.bias", "res_conv.conv.weight", "res_conv.conv.bias")

    def resnet_block(self, x_cl, sd: Dict[str, torch.Tensor], cond=None, groups=8, x1_cl=None):
        """ResnetBlock on channels-last input; ``sd`` uses the reference's key names (models.py:172-200)."""
        x = _dev32(x_cl, "x")
        B, D, H, W, c0 = x.shape
        c1 = 0 if x1_cl is None else x1_cl.shape[-1]
        cout = sd["block1.proj.conv.weight"].shape[0]
        ptrs = (_P * 12)(*[(sd[k].data_ptr() if k in sd else None) for k in self._RES_KEYS])
        y = torch.empty((B, D, H, W, cout), dtype=torch.float32, device=x.device)
        nbytes = self.lib.cd_op_scratch_bytes(B, max(c0 + c1, cout), D * H * W) + 4 * B * D * H * W * cout * 4 * 4
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _check(self.lib.cd_op_resnet_block(x.data_ptr(), c0, _ptr(x1_cl), c1, ptrs, _ptr(cond), y.data_ptr(), B, cout,
                                           _i32x3((D, H, W)), groups, ws.data_ptr(), ws.numel(), _stream()))
        return y

    _ATTN_KEYS = ("fn.norm.weight", "fn.norm.bias", "fn.fn.to_qkv.conv.weight", "fn.fn.to_out.0.conv.weight",
                  "fn.fn.to_out.0.conv.bias", "fn.fn.to_out.1.weight", "fn.fn.to_out.1.bias")

    def linear_attention(self, x_cl, sd: Dict[str, torch.Tensor]):
        """Residual(PreNorm(LinearAttention)) on channels-last input (models.py:281-329)."""
        x = _dev32(x_cl, "x")
        B, D, H, W, c = x.shape
        ptrs = (_P * 7)(*[sd[k].data_ptr() for k in self._ATTN_KEYS])
        y = torch.empty_like(x)
        nbytes = self.lib.cd_op_scratch_bytes(B, 96, D * H * W) + 4 * B * D * H * W * 96 * 4
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _check(self.lib.cd_op_linear_attention(x.data_ptr(), ptrs, y.data_ptr(), B, c, _i32x3((D, H, W)), ws.data_ptr(),
                                               ws.numel(), _stream()))
        return y
