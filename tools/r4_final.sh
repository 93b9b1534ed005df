#!/bin/bash
# Round-4 evidence of the tree as it stands: bench line (default flags), the same under rocprofv3 --stats, per-position step
# breakdown, counter passes of the timed kernels, every BASELINE configuration, the training steps.   bash tools/r4_final.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4final; mkdir -p $o gpurun_out/pmc
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 20 --warmup 5 > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err
cp $(ls $o/prof/*/*kernel_stats.csv | head -1) $o/bench_kernel_stats.csv; rm -rf $o/prof
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $o/trace -- python3 bench.py --no-train --no-cpu-baseline --no-full-chain --steps 104 --warmup 8 > $o/bench_trace.log 2>&1
python tools/step_breakdown.py $(ls $o/trace/*/*kernel_trace.csv | head -1) > $o/step_breakdown.txt; rm -rf $o/trace
bash tools/pmc_run.sh wino conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip+downsampled-diffusion_amd/csrc/conv_wino2_kernel.inc r04_wino_pmc
bash tools/pmc_run.sh wino16 conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip+downsampled-diffusion_amd/csrc/conv_wino2_kernel.inc r04_wino16_pmc
bash tools/pmc_run.sh cluster32 conv3x3_wino2_kernel downsampled-diffusion_amd/csrc/conv_wino.hip+downsampled-diffusion_amd/csrc/conv_wino2_kernel.inc r04_wino_cluster32_pmc
bash tools/pmc_run.sh convT convT_wino_kernel downsampled-diffusion_amd/csrc/conv_wino.hip+downsampled-diffusion_amd/csrc/conv_winoT_kernel.inc r04_convT_pmc
bash tools/pmc_run.sh wlocal8 conv3x3_gn_wlocal_kernel downsampled-diffusion_amd/csrc/conv_local.hip r04_wlocal8_pmc
bash tools/pmc_run.sh c32 conv3x3_c32_kernel downsampled-diffusion_amd/csrc/conv_igemm.hip+downsampled-diffusion_amd/csrc/conv_c32_kernel.inc r04_c32_pmc
bash tools/pmc_run.sh stream conv1x1_stream_kernel downsampled-diffusion_amd/csrc/conv1x1_stream.hip r04_stream_pmc
cp gpurun_out/pmc/r04_*_pmc.json profiles/            # the bench line below reads the counter summaries of THIS tree
timeout -k 10 500 python bench.py > $o/bench_latest.json 2> $o/bench_latest.err
timeout -k 10 300 python tools/sample_bench.py > $o/sample_bench.txt 2>&1
timeout -k 10 400 python tools/train_bench.py > $o/train_bench.txt 2>&1
timeout -k 10 200 python tools/encdec_bench.py > $o/encdec_bench.txt 2>&1
(make -C downsampled-diffusion_amd/csrc -j8 tune > /dev/null 2>&1 && for v in plain mo dg; do timeout -k 10 60 python tools/c32_clock.py 64 64 $v; done; timeout -k 10 60 python tools/c32_clock.py 32 64 mo; timeout -k 10 60 python tools/local_clock.py 256; timeout -k 10 60 python tools/wl_clock.py 256) 2>&1 | grep -v amdgpu > $o/clock_stamps.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tprof -- python3 tools/train_profile.py cfg3 > $o/tprof.log 2>&1
cp $(ls $o/tprof/*/*kernel_stats.csv | head -1) $o/train_cfg3_kernel_stats.csv; rm -rf $o/tprof
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $o/tg -- python3 tools/train_profile.py cfg3 graph > $o/tg.log 2>&1
python tools/train_breakdown.py $(ls $o/tg/*/*kernel_trace.csv | head -1) -v > $o/train_step_breakdown.txt; rm -rf $o/tg
head -1 $o/train_step_breakdown.txt
grep "kernel time" $o/step_breakdown.txt; grep -v amdgpu $o/sample_bench.txt $o/train_bench.txt | cut -d: -f2-
head -c 300 $o/bench_latest.json; echo
