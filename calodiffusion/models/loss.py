"""calodiffusion/models/loss.py of the reference: the training losses (load_attr("loss", name))."""
from calodiffusion_amd.loss import *  # noqa: F401,F403
from calodiffusion_amd import loss as _l

globals().update({k: v for k, v in vars(_l).items() if isinstance(v, type)})
