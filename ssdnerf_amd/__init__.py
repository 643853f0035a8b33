"""ssdnerf_amd -- MI355X-native implementation of SSDNeRF's volumetric-rendering + DDIM hot path.

Importing the package populates the ``MODULES`` / ``MODELS`` registries with the reference's type names
(``TriPlaneDecoder``, ``GaussianDiffusion``, ``DenoisingUnetMod``, ``DiffusionNeRF`` ...).  The HIP library
(``ssdnerf_amd/lib/libssdnerf_hip.so``, built by ``python -m ssdnerf_amd.build``) is loaded on first use and there is
no CPU fallback for it.
"""
from . import registry  # noqa: F401
from . import decoders, diffusion, models, unet  # noqa: F401  (register module types)
from .registry import MODELS, MODULES, build_module  # noqa: F401
