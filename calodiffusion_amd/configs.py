"""Config loading for the hot path.

The reference's config files are YAML-flow "JSON" (single quotes, trailing commas); it parses them
with ``yaml.safe_load`` (reference calodiffusion/utils/utils.py:439-443).  The same loader is used
here so that reference config files work unchanged; the files shipped in ``configs/`` carry only
the keys the denoiser reads.
"""
from __future__ import annotations

import os
from typing import Union

import yaml

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def LoadJson(file_name: str) -> dict:
    """Same name and behaviour as the reference helper (utils/utils.py:439-443)."""
    with open(file_name) as fh:
        return yaml.safe_load(fh)


def load_config(name_or_path: Union[str, dict]) -> dict:
    """Accept a dict, a path to a reference-style config, or the name of a shipped config."""
    if isinstance(name_or_path, dict):
        return name_or_path
    if os.path.exists(name_or_path):
        return LoadJson(name_or_path)
    shipped = os.path.join(_HERE, name_or_path + ".json")
    if os.path.exists(shipped):
        return LoadJson(shipped)
    raise FileNotFoundError(f"no config file or shipped config named {name_or_path!r}")
