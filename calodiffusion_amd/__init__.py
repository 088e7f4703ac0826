"""calodiffusion_amd: the CaloDiffusion denoising hot path on MI355X (gfx950).

Host side mirrors the reference's Python surface for this path
(``CaloDiffusion``, ``CondUnet``, ``DDim``/``DDPM``, ``hybrid_weight``); all arithmetic runs in the
hand-written HIP library ``lib/libcalodiff_hip.so`` reached through the C ABI declared in
``include/calodiff.h``.  There is no CPU or PyTorch fallback: every compute entry point raises if
the library or a GPU is missing.
"""
from .configs import LoadJson, load_config  # noqa: F401

__all__ = ["LoadJson", "load_config"]
