set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--steps 2 --warmup 1 --no-f32"
bash tools/gpu_ab.sh abs2_earth "--scene earth --spp 256 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_perlin "--scene two_perlin_spheres --spp 64 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_f8 "--scene final_scene --spp 8 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_f512 "--scene final_scene --spp 512 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_r1600 "--scene random_scene --width 800 --height 450 --spp 1600 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_r400 "--scene random_scene --width 800 --height 450 --spp 400 $A" nostream base j64 || exit 1
bash tools/gpu_ab.sh abs2_rl "--scene random_scene_light --width 800 --height 450 --spp 100 $A" nostream base j64 || exit 1
