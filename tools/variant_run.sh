#!/bin/bash
# usage (GPU box): tools/variant_run.sh <variant.so> <command...>  -- runs a command with the library swapped for a variant
set -e
cp libdeflate_amd/libdeflate_hip.so /tmp/ldh_keep.so
cp "$1" libdeflate_amd/libdeflate_hip.so; shift
"$@" || true
cp /tmp/ldh_keep.so libdeflate_amd/libdeflate_hip.so
