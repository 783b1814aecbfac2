"""diffsinger_amd - MI355X-native (gfx950) diffusion denoiser for DiffSinger.

Drop-in for the reference's `modules/backbones` registry and `modules/core` diffusion wrappers; the
compute path is the hand-written HIP library `libdsdenoise.so` (include/dsdenoise.h) and nothing else.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every kernel of the denoise loop
# starts with scalar loads of its ~1 KB argument block, which otherwise cross PCIe (~1 us per launch).
# Must be in the environment before the HIP runtime initialises (first CUDA/HIP call of the process).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .hparams import hparams, set_hparams  # noqa: F401,E402

__all__ = ["hparams", "set_hparams"]
