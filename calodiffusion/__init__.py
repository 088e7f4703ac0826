"""Import-path alias of the hot-path drop-in: `calodiffusion.models.*` and `calodiffusion.utils.utils` resolve to calodiffusion_amd,
so that code written against the reference's module paths (calodiffusion/models/{diffusion,calodiffusion,layerdiffusion,sample,
loss,models}.py, calodiffusion/utils/utils.py) imports the MI355X path unchanged.  Only the denoising hot path exists here: the
training / inference command-line programs, dataset IO, plotting and ControlNet of the reference are out of scope."""
