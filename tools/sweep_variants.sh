#!/bin/bash
# usage (on the GPU box): tools/sweep_variants.sh [bench args]  -- runs the phase bench once per variants/*.so
# (variants are built here with -D tuning macros; the box's copy of the tree is scratch, so the library is swapped in place)
set -e
cp libdeflate_amd/libdeflate_hip.so /tmp/ldh_keep.so
for f in variants/*.so; do
  cp "$f" libdeflate_amd/libdeflate_hip.so
  echo "== $f"
  tools/bench_phases.sh "sweep_$(basename "$f" .so).log" "$@"
done
cp /tmp/ldh_keep.so libdeflate_amd/libdeflate_hip.so
