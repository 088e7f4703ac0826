"""calodiffusion/models/diffusion.py of the reference: Diffusion (abstract base of the diffusion models)."""
from calodiffusion_amd.diffusion import *  # noqa: F401,F403
from calodiffusion_amd.diffusion import Diffusion  # noqa: F401
