set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/spp_sweep
for cfg in "final_scene 1024 1024 64" "final_scene 1024 1024 512" "random_scene 800 450 100" "random_scene 800 450 1600" "random_scene 800 450 6400"; do
  set -- $cfg
  timeout -k 10 120 python -u bench.py --scene $1 --width $2 --height $3 --spp $4 --steps 2 --warmup 1 --no-f32 --no-cpu-baseline > gpurun_out/spp_sweep/$1_$4.json 2> gpurun_out/spp_sweep/$1_$4.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/spp_sweep/$1_$4.json").read().strip().splitlines()[-1])
print("$1 $4", d["value"], d["ms_per_step"])
PY
done
