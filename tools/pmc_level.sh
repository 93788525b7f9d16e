#!/bin/bash
# usage (GPU box, repo root): tools/pmc_level.sh <tag> <level>
# bench.py at one level under rocprofv3 --pmc (two SQ passes): instruction mix and issue / wait cycles per kernel.
set -u
tag=$1; level=${2:-1}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$out/prof_${tag}_$name" -- python3 "$repo/bench.py" --level "$level" --steps 2 --warmup 1 --no-cpu-baseline --no-other-levels --no-extra-rows > "$out/${tag}_$name.log" 2>&1
  echo "$name rc=$?"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES
python3 "$repo/tools/pmc_summary.py" $(find "$out/prof_${tag}_sq1" "$out/prof_${tag}_sq2" -name '*counter_collection.csv') > "$out/${tag}_pmc_level$level.csv"
cat "$out/${tag}_pmc_level$level.csv"
