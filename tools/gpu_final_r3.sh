#!/bin/bash
# Round-3 closing session: every number DESIGN.md / README.md quote, on ONE box: bash tools/gpu_final_r3.sh
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
A="--steps 3 --warmup 1"
bash tools/gpu_r3.sh r03_fin bench scenes || exit 1
bash tools/gpu_ab.sh r03_fin_f64 "--scene final_scene --spp 64 $A" base || exit 1
bash tools/gpu_ab.sh r03_fin_f2k "--scene final_scene --width 2048 --height 2048 --spp 64 --steps 2 --warmup 1" base || exit 1
bash tools/gpu_ab.sh r03_fin_c128 "--spp 128 $A --no-f32" base || exit 1
bash tools/gpu_ab.sh r03_fin_c256 "--spp 256 $A --no-f32" base || exit 1
bash tools/gpu_ab.sh r03_fin_c512 "--spp 512 $A --no-f32" base || exit 1
bash tools/gpu_ab.sh r03_fin_s1k "--scene cornell_smoke --spp 1024 $A" base || exit 1
bash tools/gpu_ab.sh r03_fin_r1600 "--scene random_scene --width 800 --height 450 --spp 1600 --steps 2 --warmup 1" base || exit 1
bash tools/gpu_r3.sh r03_fin_det det || exit 1
bash tools/cpu_configs.sh r03_fin_cpu
