"""ddk: Python face of libddk.so (HIP kernels for the DDPM/dDDPM denoising path on MI355X)."""
from .lib import DDKError, LIB_PATH, load  # noqa: F401
