#!/bin/bash
# Round-3 profile sessions (rocprofv3 kernel trace + 4 separate PMC passes each) for the three kernels that matter:
#   r03_cornell  cornell_box 1024^2 @ 1024 spp/launch f64   (block-sorted kernel, the headline)
#   r03_random   random_scene 800x450 @ 100 spp/launch f64   (persistent kernel, fast BVH walk)
#   r03_final    final_scene 1024^2 @ 64 spp/launch f64      (persistent kernel, generic variant)
#   r03_smoke / r03_earth / r03_perlin: the other BASELINE configs (C4, C3)
# Usage through gpurun: bash tools/gpu_prof_r3.sh [cornell|random|final ...]
set -u
WHAT=${*:-cornell random final smoke earth perlin}
for w in $WHAT; do
case $w in
cornell) BENCH_ARGS="" bash tools/gpu_round.sh r03_cornell prof pmc || exit $?;;
random) BENCH_ARGS="--scene random_scene --width 800 --height 450 --spp 100" bash tools/gpu_round.sh r03_random prof pmc || exit $?;;
final) BENCH_ARGS="--scene final_scene --spp 64" bash tools/gpu_round.sh r03_final prof pmc || exit $?;;
smoke) BENCH_ARGS="--scene cornell_smoke --spp 64" bash tools/gpu_round.sh r03_smoke prof pmc || exit $?;;
earth) BENCH_ARGS="--scene earth --spp 256" bash tools/gpu_round.sh r03_earth prof pmc || exit $?;;
perlin) BENCH_ARGS="--scene two_perlin_spheres --spp 64" bash tools/gpu_round.sh r03_perlin prof pmc || exit $?;;
esac
done
