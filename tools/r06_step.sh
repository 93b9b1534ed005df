#!/bin/bash
# Round-6 quick look on one MI355X box: sampler step trace (kernel by kernel) + the bench line in the driver's form.
# usage: bash tools/r06_step.sh <tag>      -> gpurun_out/<tag>/{r06_sampler_step_breakdown.txt, bench.json}
set -e
export TMPDIR=/tmp
out=gpurun_out/${1:-r6x}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/step -- python3 bench.py --no-train --no-cpu-baseline --no-b192 --no-full-chain --steps 104 --warmup 8 > $out/step_bench.json 2> $out/step_bench.err
f=$(ls $out/step/*/*kernel_trace.csv | head -1); python3 tools/step_breakdown.py $f > $out/r06_sampler_step_breakdown.txt; rm -rf $out/step
echo step breakdown done
python3 bench.py --steps 20 --warmup 5 --no-train --no-cpu-baseline --no-b192 > $out/bench.json 2> $out/bench.err
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("images/s", d["value"], "ms/step", d["ms_per_step"], "full chain", d.get("value_full_chain"), "agree", d["config"].get("full_chain_agrees_within_3pct"))
PY
