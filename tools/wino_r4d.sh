#!/bin/bash
out=gpurun_out/r4wino; mkdir -p $out; : > $out/sweep2.txt
C=$PWD/downsampled-diffusion_amd/csrc
for lib in libddk_tune.so libddk_tune_lpf4.so libddk_tune_lpd8.so; do
  export DDK_LIB=$C/$lib
  echo "== $lib (auto variant)" >> $out/sweep2.txt
  timeout -k 10 120 python tools/wino_quick.py >> $out/sweep2.txt 2>&1 || { echo FAILED $lib; tail -5 $out/sweep2.txt; exit 1; }
  echo "== $lib stamps F 32x32 / D 16x16" >> $out/sweep2.txt
  DDK_WINO_VARIANT=2 timeout -k 10 120 python tools/wino_clock.py 32 128 128 >> $out/sweep2.txt 2>&1
  DDK_WINO_VARIANT=1 timeout -k 10 120 python tools/wino_clock.py 16 256 256 >> $out/sweep2.txt 2>&1
done
grep -v amdgpu.ids $out/sweep2.txt
