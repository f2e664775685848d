"""MI355X-native Stable Diffusion denoising path (drop-in for dawmro/pytorch_stable_diffusion's
``pipeline.generate`` / ``models[...]`` surface).  See DESIGN.md."""
__version__ = "0.1.0"

# Kernel arguments in device memory (see _native.load): must be in the environment before the HIP runtime initialises, so it is
# set as early as this package can -- at import; an explicit setting of the variable wins.
import os as _os
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
