#!/bin/bash
# Round-6 evidence on one MI355X box: sampler step trace, bench under the profiler, counter passes of the level chain (inside the whole
# UNet forward) and of the kernels whose source changed this round, the captured cfg3 training step, the bench line in the driver's form.
# Outputs under gpurun_out/r6ev/ (copy the summaries into profiles/).
set -e
export TMPDIR=/tmp
out=gpurun_out/r6ev
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/step -- python3 bench.py --no-train --no-cpu-baseline --no-b192 --no-full-chain --steps 104 --warmup 8 > $out/step_bench.json 2> $out/step_bench.err
f=$(ls $out/step/*/*kernel_trace.csv | head -1); python3 tools/step_breakdown.py $f > $out/r06_sampler_step_breakdown.txt; rm -rf $out/step
echo step breakdown done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 20 --warmup 5 > $out/r06_bench_under_rocprof.json 2> $out/bench_rocprof.err
f=$(ls $out/bench/*/*kernel_stats.csv | head -1); cp $f $out/r06_bench_kernel_stats.csv; rm -rf $out/bench
echo bench under rocprof done
bash tools/pmc_run.sh unet level_chain_kernel downsampled-diffusion_amd/csrc/level_chain.hip r06_chain4_pmc > $out/pmc_chain4.log 2>&1
echo pmc chain done
bash tools/pmc_run.sh wlocal8 conv3x3_gn_wlocal_kernel downsampled-diffusion_amd/csrc/conv_local.hip r06_wlocal8_pmc > $out/pmc_wlocal8.log 2>&1
bash tools/pmc_run.sh local4 conv3x3_gn_local_kernel downsampled-diffusion_amd/csrc/conv_local.hip r06_local4_pmc > $out/pmc_local4.log 2>&1
bash tools/pmc_run.sh kvctx attn_kvctx_kernel downsampled-diffusion_amd/csrc/attention.hip r06_kvctx_pmc > $out/pmc_kvctx.log 2>&1
echo pmc done
rocprofv3 --kernel-trace --output-format csv -d $out/train -- python3 tools/train_profile.py cfg3 merged > $out/train_profile.log 2>&1
f=$(ls $out/train/*/*kernel_trace.csv | head -1); python3 tools/train_breakdown.py $f > $out/r06_train_step_breakdown.txt; rm -rf $out/train
echo train breakdown done
python3 bench.py --steps 20 --warmup 5 > $out/r06_bench_driver_form.json 2> $out/bench_driver_form.err
python3 bench.py > $out/r06_bench_latest.json 2> $out/bench_latest.err
echo bench done
