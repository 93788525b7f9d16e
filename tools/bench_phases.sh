#!/bin/bash
# usage: tools/bench_phases.sh <logname> [bench args...]  -> prints value + per-phase ms
log=$1; shift
timeout -k 10 300 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-levels --no-extra-rows "$@" > gpurun_out/$log 2>&1 || echo "bench exit nonzero (verification off?)"
python3 - "$log" <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/"+sys.argv[1]) if l.startswith("{")][0])
print("%.3f GB/s verified=%s"%(d["value"],d["config"]["verified"]), {k:round(v,2) for k,v in d["config"]["phase_ms_per_step"].items()})
PY
