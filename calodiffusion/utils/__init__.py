"""calodiffusion.utils: the reference's module names over calodiffusion_amd (see calodiffusion/__init__.py)."""
