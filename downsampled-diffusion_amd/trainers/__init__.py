"""Training harness for the HIP-backed DDPM/dDDPM (counterpart of the reference's trainers/ package)."""
