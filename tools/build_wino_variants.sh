#!/bin/bash
# Builds tuning-library variants that differ only in conv_wino.hip's compile-time knobs:  tools/build_wino_variants.sh "name:-Dflags" ...
# -> downsampled-diffusion_amd/csrc/libddk_tune_<name>.so  (run after `make -C downsampled-diffusion_amd/csrc tune`)
set -e
cd "$(dirname "$0")/../downsampled-diffusion_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function -Wno-inline-asm -DDDK_TUNING"
others=$(ls *.tune.o | grep -v conv_wino.tune.o)
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c conv_wino.hip -o conv_wino.$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $others conv_wino.$name.o -o libddk_tune_$name.so
  echo built libddk_tune_$name.so
done
