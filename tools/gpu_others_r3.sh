#!/bin/bash
# The three scenes of scenes.rs outside the BASELINE configs (simple_light, two_checker_spheres, random_scene_light) at 1024^2 @ 64 spp
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r03_others; mkdir -p $OUT
for sc in simple_light two_checker_spheres random_scene_light; do
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --scene $sc --width 1024 --height 1024 --spp-per-step 64 --steps 2 --warmup 1 > $OUT/$sc.json 2> $OUT/$sc.err; rc=$?
  echo "$sc rc=$rc $(python -c "import json; d=json.load(open('$OUT/$sc.json')); print(d['value'], d.get('f32_value'), d['ms_per_step'], d['casts_per_sample'])" 2>/dev/null)" | tee -a $OUT/summary.txt
  case $rc in 124|137|134|139) exit $rc;; esac
done
