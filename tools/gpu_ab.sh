#!/bin/bash
# A/B timing of variant libraries on the GPU box: bash tools/gpu_ab.sh <tag> "<bench args>" lib1 lib2 ...
# (lib = name under ray-tracing_amd/ab/, or "base" for the product library)
set -u
TAG=$1; ARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd "$REPO"
for lib in "$@"; do
  if [ "$lib" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$REPO/ray-tracing_amd/ab/librt_hip_$lib.so; fi
  timeout -k 10 300 python -u bench.py --no-cpu-baseline $ARGS > "$OUT/$lib.json" 2> "$OUT/$lib.err"; rc=$?
  echo "$lib rc=$rc $(python -c "import json,sys; d=json.load(open('$OUT/$lib.json')); print(d['value'], d.get('f32_value'), d['roofline']['launch_ms'])" 2>/dev/null)" | tee -a "$OUT/summary.txt"
  case $rc in 124|137|134|139) exit $rc;; esac
done
