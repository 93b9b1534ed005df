"""Model names accepted by the CLI (reference models/config.py:1)."""
MODEL_NAMES = ['ddpm']
