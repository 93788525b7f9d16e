#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Runs bench.py under rocprofv3: one --kernel-trace --stats run and three separate --pmc passes
# (HBM fetch, HBM write, SQ mix); leaves small summaries in gpurun_out/<tag>_*.csv for profiles/.
set -u
tag=$1
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { # name, timeout, rocprof args...
  local name=$1 to=$2; shift 2
  timeout -s KILL "$to" rocprofv3 "$@" --output-format csv -d "$out/prof_${tag}_$name" -- python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-levels --no-extra-rows > "$out/${tag}_$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  return $rc
}
run stats 300 --kernel-trace --stats || exit 1
f=$(find "$out/prof_${tag}_stats" -name '*kernel_stats.csv' | head -1)
python3 "$repo/tools/prof_summary.py" "$f" > "$out/${tag}_kernel_stats.csv" && cat "$out/${tag}_kernel_stats.csv"
run fetch 300 --pmc FETCH_SIZE || exit 1
run write 300 --pmc WRITE_SIZE || exit 1
run sq 300 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD || exit 1
python3 "$repo/tools/pmc_summary.py" $(find "$out/prof_${tag}_fetch" "$out/prof_${tag}_write" "$out/prof_${tag}_sq" -name '*counter_collection.csv') > "$out/${tag}_pmc.csv" && cat "$out/${tag}_pmc.csv"
