set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
A="--steps 2 --warmup 1 --no-f32"
bash tools/gpu_ab.sh abs_f64 "--scene final_scene --spp 64 $A" nostream base j64 j128 nostream base || exit 1
bash tools/gpu_ab.sh abs_f512 "--scene final_scene --spp 512 $A" nostream base || exit 1
bash tools/gpu_ab.sh abs_r100 "--scene random_scene --width 800 --height 450 --spp 100 $A" nostream base j64 j128 nostream base || exit 1
bash tools/gpu_ab.sh abs_r1600 "--scene random_scene --width 800 --height 450 --spp 1600 $A" nostream base || exit 1
bash tools/gpu_ab.sh abs_f2k "--scene final_scene --width 2048 --height 2048 --spp 64 $A" nostream base j64 || exit 1
