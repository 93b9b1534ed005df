from .wrapper import get_downsampling, get_upsampling
