#!/bin/bash
# Throughput of every BASELINE config scene on one GPU: bash tools/gpu_scenes.sh <tag> [lib]
set -u
TAG=$1; LIB=${2:-base}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$REPO"
[ "$LIB" = base ] || export RT_HIP_LIB=$REPO/ray-tracing_amd/ab/librt_hip_$LIB.so
run() { # name args...
  local name=$1; shift
  timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; rc=$?
  echo "$name rc=$rc $(python -c "import json; d=json.load(open('$OUT/$name.json')); print(d['value'], d.get('f32_value'), d['roofline']['launch_ms'], d['casts_per_sample'], d['nan_retries'])" 2>/dev/null)" | tee -a "$OUT/summary.txt"
  case $rc in 124|137|134|139) exit $rc;; esac
}
run cornell_box --steps 2 --warmup 1
run random_scene --scene random_scene --width 800 --height 450 --spp-per-step 100 --steps 2 --warmup 1
run cornell_smoke --scene cornell_smoke --spp-per-step 64 --steps 2 --warmup 1
run earth --scene earth --spp-per-step 256 --steps 2 --warmup 1
run two_perlin_spheres --scene two_perlin_spheres --spp-per-step 64 --steps 2 --warmup 1
run final_scene --scene final_scene --spp-per-step 8 --steps 2 --warmup 1
if [ "${NOSORT:-0}" = 1 ]; then
  run earth_nosort --scene earth --spp-per-step 256 --steps 2 --warmup 1 --no-sort
  run two_perlin_nosort --scene two_perlin_spheres --spp-per-step 64 --steps 2 --warmup 1 --no-sort
fi
