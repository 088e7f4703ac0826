"""calodiffusion/models/sample.py of the reference: the samplers (load_attr("sampler", name) resolves them by class name)."""
from calodiffusion_amd.sample import *  # noqa: F401,F403
from calodiffusion_amd import sample as _s

globals().update({k: v for k, v in vars(_s).items() if isinstance(v, type)})
