#!/bin/bash
# Round-3 soak of the generated-scene parity checks on the final build: bash tools/gpu_soak_r3.sh <first> <count-per-generator>
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-20000}; N=${2:-150}
OUT=gpurun_out/r03_soak; mkdir -p $OUT
for g in random_scene cross_tie_scene tie_scene medium_scene instancing_scene texture_scene extreme_scene far_scene pile_scene odd_lights_scene; do
  timeout -k 10 900 python -u tools/gpu_fuzz.py $F $((F + N)) $g > $OUT/$g.log 2>&1; rc=$?
  echo "$g rc=$rc $(tail -1 $OUT/$g.log)" | tee -a $OUT/summary.txt
  case $rc in 124|137|134|139) exit $rc;; esac
done
