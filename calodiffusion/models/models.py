"""calodiffusion/models/models.py of the reference: the networks of the hot path.  CondUnet / ResNet here are parameter containers
with the reference's state_dict keys; their arithmetic runs in the HIP library (calodiffusion_amd/engine.py)."""
from calodiffusion_amd.unet import CondUnet  # noqa: F401
from calodiffusion_amd.resnet import ResDense, ResNet  # noqa: F401
