set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/gpu_ab.sh abs3_earth "--scene earth --spp 256 --steps 6 --warmup 2" nostream base nostream base || exit 1
bash tools/gpu_ab.sh abs3_earth1k "--scene earth --spp 1024 --steps 3 --warmup 1" nostream base || exit 1
