#!/bin/bash
# Round-4 evidence pass: counters of the Winograd kernels + per-position step breakdown.   bash tools/r4_profile.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4prof gpurun_out/pmc
bash tools/pmc_run.sh wino conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip r04_wino_pmc
bash tools/pmc_run.sh wino16 conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip r04_wino16_pmc
bash tools/pmc_run.sh cluster32 conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip r04_wino_cluster32_pmc
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4prof/trace -- python3 bench.py --no-train --no-cpu-baseline --no-full-chain --steps 104 --warmup 8 > gpurun_out/r4prof/bench_trace.log 2>&1
python tools/step_breakdown.py $(ls gpurun_out/r4prof/trace/*/*kernel_trace.csv | head -1) > gpurun_out/r4prof/step_breakdown.txt
rm -rf gpurun_out/r4prof/trace
grep "kernel time" gpurun_out/r4prof/step_breakdown.txt
python - <<'PY'
import json
for t in ("r04_wino_pmc", "r04_wino16_pmc", "r04_wino_cluster32_pmc"):
    d = json.load(open(f"gpurun_out/pmc/{t}.json"))
    print(t, "busy", round(d.get("mfma_busy", 0), 3), "dur", round(d.get("duration_us_unprofiled_trace", 0), 2), "traffic MB", round(d.get("traffic_bytes_per_launch", 0) / 1e6, 2), "l2", round(d.get("l2_hit", 0), 3))
PY
