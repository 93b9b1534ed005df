#!/bin/bash
# Counter passes of one kernel on the GPU box:  tools/pmc_run.sh <case> <kernel substring> <source file> <out tag>
# Separate rocprofv3 runs per counter group (SQ block | FETCH_SIZE | WRITE_SIZE | L2 / L1->L2 requests) plus one trace-only run for
# the un-profiled duration; summaries land in gpurun_out/pmc/<tag>.json (copy into profiles/).
set -e
case=$1; kname=$2; src=$3; tag=$4
export TMPDIR=/tmp
out=gpurun_out/pmc/$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $out/sq -- python3 tools/pmc_one.py $case > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/pmc_one.py $case > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/pmc_one.py $case > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -- python3 tools/pmc_one.py $case > $out/l2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/tcp -- python3 tools/pmc_one.py $case > $out/tcp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/pmc_one.py $case > $out/trace.log 2>&1
python3 tools/pmc_summary.py "$kname" $src gpurun_out/pmc/$tag.json $out/sq $out/fetch $out/write $out/l2 $out/tcp $out/trace
# keep the raw counter CSVs small: only the kernel's rows
for d in sq fetch write l2 tcp; do
  f=$(ls $out/$d/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "$kname" $f) > gpurun_out/pmc/${tag}_$d.csv
done
f=$(ls $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/pmc/${tag}_kernel_stats.csv
rm -rf $out
