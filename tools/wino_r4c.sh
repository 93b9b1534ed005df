#!/bin/bash
# Stagger / DMA-split sweep of the 8-matrix-wave Winograd kernel: libraries built by tools/build_wino_variants.sh
out=gpurun_out/r4wino; mkdir -p $out; : > $out/sweep.txt
C=$PWD/downsampled-diffusion_amd/csrc
for lib in libddk_tune.so libddk_tune_nostg.so libddk_tune_lp8.so libddk_tune_lp8nostg.so; do
  export DDK_LIB=$C/$lib
  for v in 1 2; do
    echo "== $lib variant $v" >> $out/sweep.txt
    DDK_WINO_VARIANT=$v timeout -k 10 120 python tools/wino_quick.py >> $out/sweep.txt 2>&1 || { echo FAILED $lib $v; tail -5 $out/sweep.txt; exit 1; }
  done
  echo "== $lib stamps F 32x32 / D 16x16" >> $out/sweep.txt
  DDK_WINO_VARIANT=2 timeout -k 10 120 python tools/wino_clock.py 32 128 128 >> $out/sweep.txt 2>&1
  DDK_WINO_VARIANT=1 timeout -k 10 120 python tools/wino_clock.py 16 256 256 >> $out/sweep.txt 2>&1
done
export DDK_LIB=$C/libddk_tune.so
echo "== timeline, staggered, F" >> $out/sweep.txt
DDK_WINO_VARIANT=2 DDK_WINO_DEBUG=17 timeout -k 10 120 python tools/wino_clock.py 32 128 128 >> $out/sweep.txt 2>&1
grep -v amdgpu.ids $out/sweep.txt
