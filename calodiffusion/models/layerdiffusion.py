"""calodiffusion/models/layerdiffusion.py of the reference: LayerDiffusion."""
from calodiffusion_amd.layerdiffusion import *  # noqa: F401,F403
from calodiffusion_amd.layerdiffusion import LayerDiffusion  # noqa: F401
