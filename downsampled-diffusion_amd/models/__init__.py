"""Drop-in ``models`` package: same names as the reference's models/__init__.py:1-6, HIP-backed."""
from .config import MODEL_NAMES
from .unet.unet import Unet
from .diffusion.ddpm import DDPM
from .diffusion.dddpm import DownsampleDDPM, DownsampleDDPMAutoencoder
from .downsampled.wrapper import get_downsampling, get_upsampling

__all__ = ["MODEL_NAMES", "Unet", "DDPM", "DownsampleDDPM", "DownsampleDDPMAutoencoder",
           "get_downsampling", "get_upsampling"]
