#!/bin/bash
# Round-4 comparison of the Winograd kernels inside one GPU call (tuning build: DDK_WINO_VARIANT 0 = 4+4 kernel, 1 = 8-matrix-wave
# kernel with 64-channel tiles everywhere, 2 = 128-channel tiles where N allows, unset = the production choice).
export DDK_LIB=$PWD/downsampled-diffusion_amd/csrc/libddk_tune.so
out=gpurun_out/r4wino; mkdir -p $out
for v in 0 1 2 auto; do
  if [ $v = auto ]; then unset DDK_WINO_VARIANT; else export DDK_WINO_VARIANT=$v; fi
  echo "== variant $v" | tee -a $out/quick.txt
  timeout -k 10 120 python tools/wino_quick.py >> $out/quick.txt 2>&1 || { echo "variant $v FAILED"; tail -5 $out/quick.txt; exit 1; }
done
cat $out/quick.txt
for v in 0 2; do
  echo "== stamps, variant $v, 32x32 128->128" >> $out/clock.txt
  DDK_WINO_VARIANT=$v timeout -k 10 120 python tools/wino_clock.py 32 128 128 >> $out/clock.txt 2>&1
done
for v in 0 1; do
  echo "== stamps, variant $v, 16x16 256->256" >> $out/clock.txt
  DDK_WINO_VARIANT=$v timeout -k 10 120 python tools/wino_clock.py 16 256 256 >> $out/clock.txt 2>&1
done
cat $out/clock.txt
