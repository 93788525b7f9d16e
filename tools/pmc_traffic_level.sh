#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic_level.sh <tag> <level>
# bench.py at one level under rocprofv3 --pmc, one pass per counter (FETCH_SIZE; WRITE_SIZE): HBM-side bytes per kernel.
set -u
tag=$1; level=${2:-1}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() { # name counter
  timeout -s KILL 180 rocprofv3 --pmc "$2" --output-format csv -d "$out/prof_${tag}_$1" -- python3 "$repo/bench.py" --level "$level" --steps 2 --warmup 1 --no-cpu-baseline --no-other-levels --no-extra-rows > "$out/${tag}_$1.log" 2>&1
  local rc=$?; echo "$1 rc=$rc"; return $rc
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
python3 "$repo/tools/pmc_summary.py" $(find "$out/prof_${tag}_fetch" "$out/prof_${tag}_write" -name '*counter_collection.csv') > "$out/${tag}_pmc_traffic_level$level.csv"
python3 "$repo/tools/pmc_traffic.py" "$out/${tag}_pmc_traffic_level$level.csv" "$out/${tag}_pmc_traffic_level$level.json" "8192x131072 level $level"
