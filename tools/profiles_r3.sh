#!/bin/bash
# After `gpurun -- bash tools/gpu_prof_r3.sh`: summaries of every session into profiles/r03_* (+ pmc_views.json keys bench.py looks up)
set -e
cd "$(dirname "$0")/.."
s() { [ -d gpurun_out/r03_$1 ] && python tools/pmc_summary.py gpurun_out/r03_$1 profiles/r03_$1 render_kernel "$2" > /dev/null && echo "profiles/r03_$1 <- $2"; }
s cornell cornell_box_1024x1024_1024spp_f64
s random random_scene_800x450_100spp_f64
s final final_scene_1024x1024_64spp_f64
s smoke cornell_smoke_1024x1024_64spp_f64
s earth earth_1024x1024_256spp_f64
s perlin two_perlin_spheres_1024x1024_64spp_f64
