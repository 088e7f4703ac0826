"""calodiffusion/utils/utils.py of the reference, as far as the hot path uses it: device choice, coordinate images, load_attr,
ReverseNorm."""
from calodiffusion_amd.utils import *  # noqa: F401,F403
from calodiffusion_amd.utils import create_phi_image, create_R_Z_image, get_device, load_attr, subsample_alphas  # noqa: F401
from calodiffusion_amd.postprocess import ReverseNorm, ReverseNormCaloChall  # noqa: F401
