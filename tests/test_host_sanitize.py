"""SURVEY.md section 5 (sanitizers): the HOST half of libddk -- plan builder, kernel choosers, workspace carve-up, weight-slot
packing, sampler bookkeeping -- under AddressSanitizer + UndefinedBehaviorSanitizer on this GPU-less machine.  `make asan` compiles
every translation unit as usual with the host side instrumented and every kernel launch / runtime call rerouted to checkers
(csrc/host_sanitize.h): nothing runs on a device, but every launch's geometry is validated and every pointer it would hand to the
GPU -- including the pointer words inside by-value parameter blocks, and base + extent for the conv and GroupNorm launchers --
must lie inside an arena of exactly the size the library's own size queries returned.  tests/host/plan_walk.cpp walks all five
BASELINE configurations (packing, forward with and without the in-launch GroupNorm, a three-step eager sampler chain) plus
off-config widths / depths / odd batches."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "downsampled-diffusion_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_half_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1)), "asan"], capture_output=True, text=True, timeout=1500)
    assert build.returncode == 0, (build.stdout + build.stderr)[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([os.path.join(CSRC, "host_walk_asan")], capture_output=True, text=True, timeout=600, env=env)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    last = run.stdout.strip().splitlines()[-1]
    assert last.startswith("host walk:") and " 0 pointer/geometry errors, 0 failed expectations" in last, last
    assert int(last.split()[2]) > 5000          # every launch of 8 plans x (packing + 3 forwards + 3 reverse steps) went through the checker
