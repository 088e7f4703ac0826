"""calodiffusion/models/calodiffusion.py of the reference: CaloDiffusion."""
from calodiffusion_amd.calodiffusion import *  # noqa: F401,F403
from calodiffusion_amd.calodiffusion import CaloDiffusion  # noqa: F401
