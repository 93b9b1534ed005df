#!/bin/bash
# Ablations + per-wave barrier timeline of the 8-matrix-wave Winograd kernel (tuning build).
export DDK_LIB=$PWD/downsampled-diffusion_amd/csrc/libddk_tune.so
out=gpurun_out/r4wino; mkdir -p $out; : > $out/ablate.txt
for shape in "32 128 128 2" "16 256 256 1"; do
  set -- $shape
  for dbg in 1 7 9 15 17; do
    echo "== variant $4, ${1}x$1 $2->$3, DDK_WINO_DEBUG=$dbg (1 stamps, +2 no loader DMA, +4 no matrix DMA, +8 no V work, 17 timeline)" >> $out/ablate.txt
    DDK_WINO_VARIANT=$4 DDK_WINO_DEBUG=$dbg timeout -k 10 120 python tools/wino_clock.py $1 $2 $3 >> $out/ablate.txt 2>&1 || { echo FAILED; tail -5 $out/ablate.txt; exit 1; }
  done
done
grep -v amdgpu.ids $out/ablate.txt
