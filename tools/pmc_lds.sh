#!/bin/bash
# usage (GPU box, repo root): tools/pmc_lds.sh <tag>  -- one --pmc pass with the LDS / VALU activity counters
set -u
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d "$out/prof_${tag}_lds" -- python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-levels --no-extra-rows > "$out/${tag}_lds.log" 2>&1
echo "rc=$?"
python3 "$repo/tools/pmc_summary.py" $(find "$out/prof_${tag}_lds" -name '*counter_collection.csv') > "$out/${tag}_lds.csv" && cat "$out/${tag}_lds.csv"
