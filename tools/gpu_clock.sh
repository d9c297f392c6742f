#!/bin/bash
# Phase-clock profile (RT_PHASE_CLOCK build, tools/build_ab.py clock:RT_PHASE_CLOCK): bash tools/gpu_clock.sh <tag> <bench args...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$REPO"
export RT_HIP_LIB=$REPO/ray-tracing_amd/ab/librt_hip_clock.so
timeout -k 10 280 python -u bench.py --steps 1 --warmup 0 --no-f32 --no-cpu-baseline "$@" > "$OUT/out.txt" 2>&1; rc=$?
echo "rc=$rc"; grep -c PC "$OUT/out.txt"
exit $rc
