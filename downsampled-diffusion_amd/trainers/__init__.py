"""Training harness for the HIP-backed DDPM/dDDPM (counterpart of the reference's trainers/ package)."""
from .wrapper import setup_trainer
from .trainer_ddpm import TrainerDDPM, TrainerDownsampleDDPM
from .ema import EMA
