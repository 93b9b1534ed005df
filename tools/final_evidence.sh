set -e
mkdir -p gpurun_out/r3final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > gpurun_out/r3final/bench_latest.json 2> gpurun_out/r3final/bench_latest.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3final/prof -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3final/bench_under_rocprof.json 2> gpurun_out/r3final/bench_under_rocprof.err
cp $(ls gpurun_out/r3final/prof/*/*kernel_stats.csv | head -1) gpurun_out/r3final/bench_kernel_stats.csv
rm -rf gpurun_out/r3final/prof
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3final/trace -- python3 bench.py --no-train --no-cpu-baseline --steps 104 --warmup 8 > gpurun_out/r3final/bench_trace.log 2>&1
python tools/step_breakdown.py $(ls gpurun_out/r3final/trace/*/*kernel_trace.csv | head -1) > gpurun_out/r3final/step_breakdown.txt
rm -rf gpurun_out/r3final/trace
timeout -k 10 300 python tools/sample_bench.py > gpurun_out/r3final/sample_bench.txt 2>&1
timeout -k 10 300 python tools/train_bench.py > gpurun_out/r3final/train_bench.txt 2>&1
timeout -k 10 200 python tools/edge_bench.py > gpurun_out/r3final/edge_bench.txt 2>&1
timeout -k 10 200 python tools/ws_bench.py 32 probe > gpurun_out/r3final/ws_bench.txt 2>&1
tail -3 gpurun_out/r3final/step_breakdown.txt | head -1; grep "kernel time" gpurun_out/r3final/step_breakdown.txt; cat gpurun_out/r3final/sample_bench.txt gpurun_out/r3final/train_bench.txt | grep -v amdgpu
head -c 400 gpurun_out/r3final/bench_latest.json
