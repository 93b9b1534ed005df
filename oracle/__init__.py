"""CPU oracle for the DDPM/dDDPM denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU / numpy functional code, the algorithm of
the reference (simonamtoft/downsampled-diffusion) for the path named in
BASELINE.json: the epsilon-prediction UNet forward, the noise-schedule
arithmetic, the T-step p_sample loop, the dDDPM resampler networks and the
training-step arithmetic.  Every function cites the reference file:line it follows.

Pinning: the restatement is checked (tests/test_oracle_golden.py) against golden
vectors in tests/golden/ that were produced by importing the reference itself in
the build container (tools/gen_golden.py, committed).  The reference ships no
tests or fixtures of its own, so those vectors are the pin.

Rules: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import anything from here, and only as the checker / timed CPU baseline.  The
product path (downsampled-diffusion_amd/) never imports oracle/ and has no CPU
fallback: it raises if the HIP library is missing.
"""
