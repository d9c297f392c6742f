#!/bin/bash
# Round-2 profile sessions (rocprofv3 kernel trace + 4 separate PMC passes each) for the three kernels that matter:
#   r02_cornell  cornell_box 1024^2 @ 1024 spp/launch f64   (block-sorted kernel, the headline)
#   r02_random   random_scene 800x450 @ 100 spp/launch f64   (persistent kernel, fast BVH walk)
#   r02_final    final_scene 1024^2 @ 64 spp/launch f64      (persistent kernel, generic variant)
#   r02_smoke / r02_earth / r02_perlin: the other BASELINE configs (C4, C3)
# Usage through gpurun: bash tools/gpu_prof_r2.sh [cornell|random|final ...]
set -u
WHAT=${*:-cornell random final smoke earth perlin}
for w in $WHAT; do
case $w in
cornell) BENCH_ARGS="" bash tools/gpu_round.sh r02_cornell prof pmc || exit $?;;
random) BENCH_ARGS="--scene random_scene --width 800 --height 450 --spp 100" bash tools/gpu_round.sh r02_random prof pmc || exit $?;;
final) BENCH_ARGS="--scene final_scene --spp 64" bash tools/gpu_round.sh r02_final prof pmc || exit $?;;
smoke) BENCH_ARGS="--scene cornell_smoke --spp 64" bash tools/gpu_round.sh r02_smoke prof pmc || exit $?;;
earth) BENCH_ARGS="--scene earth --spp 256" bash tools/gpu_round.sh r02_earth prof pmc || exit $?;;
perlin) BENCH_ARGS="--scene two_perlin_spheres --spp 64" bash tools/gpu_round.sh r02_perlin prof pmc || exit $?;;
esac
done
