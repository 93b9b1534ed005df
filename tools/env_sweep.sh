# HIP runtime knobs against the sampler step (profiles/r04_runtime_knobs.txt): bash tools/env_sweep.sh
run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --no-train --no-cpu-baseline --no-full-chain --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run AMD_OPT_FLUSH=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run A=2
