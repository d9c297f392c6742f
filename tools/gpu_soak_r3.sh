#!/bin/bash
# Round-3 soak of the generated-scene checks on the final build: bash tools/gpu_soak_r3.sh <first> <count-per-generator> [kernels-count]
#   gpu_fuzz.py over all ten generators (GPU against the oracle), then — with a third argument — gpu_fuzz_kernels.py (the job machinery
#   of both render kernels at ~0.7 Mpixel frames) and gpu_f32_sanity.py over a few generators
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-20000}; N=${2:-150}; K=${3:-0}
OUT=gpurun_out/r03_soak; mkdir -p $OUT
for g in random_scene cross_tie_scene tie_scene medium_scene instancing_scene texture_scene extreme_scene far_scene pile_scene odd_lights_scene; do
  timeout -k 10 900 python -u tools/gpu_fuzz.py $F $((F + N)) $g > $OUT/$g.log 2>&1; rc=$?
  echo "$g rc=$rc $(tail -1 $OUT/$g.log)" | tee -a $OUT/summary.txt
  case $rc in 124|137|134|139) exit $rc;; esac
done
[ "$K" -gt 0 ] || exit 0
for g in random_scene medium_scene instancing_scene odd_lights_scene; do
  timeout -k 10 900 python -u tools/gpu_fuzz_kernels.py $F $((F + K)) $g > $OUT/kernels_$g.log 2>&1; rc=$?
  echo "kernels $g rc=$rc $(tail -1 $OUT/kernels_$g.log)" | tee -a $OUT/summary.txt
  case $rc in 124|137|134|139) exit $rc;; esac
done
timeout -k 10 900 python -u tools/gpu_f32_sanity.py $F $((F + K)) random_scene medium_scene texture_scene extreme_scene > $OUT/f32.log 2>&1; rc=$?
echo "f32 sanity rc=$rc $(tail -1 $OUT/f32.log)" | tee -a $OUT/summary.txt
exit $rc
