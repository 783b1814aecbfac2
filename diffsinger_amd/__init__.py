"""diffsinger_amd - MI355X-native (gfx950) diffusion denoiser for DiffSinger.

Drop-in for the reference's `modules/backbones` registry and `modules/core` diffusion wrappers; the
compute path is the hand-written HIP library `libdsdenoise.so` (include/dsdenoise.h) and nothing else.
"""
from .hparams import hparams, set_hparams  # noqa: F401

__all__ = ["hparams", "set_hparams"]
