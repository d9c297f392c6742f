#!/bin/bash
# A/B of the LDS texel cache experiment (-DRT_TEX_LDS_CACHE): parity of the cached path, then throughput on earth and final_scene.
set -u
TAG=${1:-r02_tex}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$REPO"
for lib in tex4k tex512; do
  RT_HIP_LIB=$REPO/ray-tracing_amd/ab/librt_hip_$lib.so timeout -k 10 400 python -u -m pytest -x -q --timeout 240 --timeout-method thread tests/test_gpu_parity.py -k "earth or final_scene" > "$OUT/parity_$lib.log" 2>&1; rc=$?
  echo "parity $lib rc=$rc $(tail -1 "$OUT/parity_$lib.log")" | tee -a "$OUT/summary.txt"
  [ $rc = 0 ] || exit $rc
done
bash tools/gpu_ab.sh ${TAG}_earth "--steps 3 --warmup 1 --scene earth --spp 256" base tex4k base tex4k || exit $?
bash tools/gpu_ab.sh ${TAG}_final "--steps 2 --warmup 1 --scene final_scene --spp 16" ctl1024 tex512 ctl1024 tex512 || exit $?
