#!/bin/bash
# sorted vs persistent wave kernel on the no-BVH scenes: bash tools/gpu_nosort.sh <tag>
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$REPO"
run() { # name args...
  local name=$1; shift
  timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; rc=$?
  echo "$name rc=$rc $(python -c "import json; d=json.load(open('$OUT/$name.json')); print(d['value'], d.get('f32_value'), d['roofline']['launch_ms'], d['casts_per_sample'], d['nan_retries'])" 2>/dev/null)" | tee -a "$OUT/summary.txt"
  case $rc in 124|137|134|139) exit $rc;; esac
}
for sc in cornell_box cornell_smoke simple_light two_checker_spheres; do
  run ${sc}_sort --scene $sc --spp-per-step 64 --steps 2 --warmup 1
  run ${sc}_nosort --scene $sc --spp-per-step 64 --steps 2 --warmup 1 --no-sort
done
