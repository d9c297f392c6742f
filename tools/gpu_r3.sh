#!/bin/bash
# Round-3 GPU session: new tests first (fail fast), the whole GPU suite, bench, the cost of the reproducible mode, launcher under torchrun.
# Usage through gpurun:  bash tools/gpu_r3.sh <tag> [new|all|bench|benchq|det|torchrun|scenes|main1 ...]
set -u
TAG=${1:-r03}; shift || true
WHAT=${*:-new all bench}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
fatal() { case $1 in 124|137|134|139|-6|-11) echo "step '$2' ended with $1: stopping" | tee -a "$OUT/status.txt"; exit "$1";; esac; }
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/status.txt" >&2
  timeout -k 10 "$secs" "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/status.txt" >&2
  fatal $rc "$name"
  return $rc
}
bench1() {  # bench1 <name> <bench args...>: one short bench line -> summary
  local name=$1; shift
  step "bench_$name" 300 python -u bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -20 "$OUT/$name.err"; return 1; }
  echo "$name $(python -c "import json; d=json.load(open('$OUT/$name.json')); print(d['value'], d.get('f32_value'), d['roofline']['launch_ms'], d['casts_per_sample'])" 2>/dev/null)" | tee -a "$OUT/summary.txt"
}
cd "$REPO"
PYT="python -u -m pytest -x -v --timeout 240 --timeout-method thread"
for w in $WHAT; do
case $w in
new) step pytest_new 600 $PYT tests/test_gpu_round3.py tests/test_zz_no_alloc.py > "$OUT/pytest_new.log" 2>&1 || { tail -40 "$OUT/pytest_new.log"; exit 1; }; tail -3 "$OUT/pytest_new.log";;
all) step pytest_gpu 1100 $PYT tests -m gpu > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }; tail -3 "$OUT/pytest_gpu.log"
     cp gpurun_out/pixel_allowance.jsonl "$OUT/" 2>/dev/null;;
bench) step bench 600 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }; cat "$OUT/bench.json";;
benchq) step benchq 600 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/benchq.json" 2> "$OUT/benchq.err" || { tail -20 "$OUT/benchq.err"; exit 1; }; cat "$OUT/benchq.json";;
scenes) step scenes 1100 bash tools/gpu_scenes.sh $TAG > "$OUT/scenes.txt" 2> "$OUT/scenes.err" || { tail -20 "$OUT/scenes.err"; exit 1; }; cat "$OUT/scenes.txt";;
det)  # what RT_RENDER_DETERMINISTIC costs: default / wave kernel / deterministic on the same box
  bench1 cornell_default --no-f32 || exit 1
  bench1 cornell_nosort --no-f32 --no-sort || exit 1
  bench1 cornell_det --no-f32 --deterministic || exit 1
  bench1 cornell128_default --no-f32 --spp 128 || exit 1
  bench1 cornell128_det --no-f32 --spp 128 --deterministic || exit 1
  bench1 random_default --no-f32 --scene random_scene --width 800 --height 450 --spp 100 || exit 1
  bench1 random_det --no-f32 --scene random_scene --width 800 --height 450 --spp 100 --deterministic || exit 1
  bench1 final_default --no-f32 --scene final_scene --spp 64 || exit 1
  bench1 final_det --no-f32 --scene final_scene --spp 64 --deterministic || exit 1
  bench1 earth_default --no-f32 --scene earth --spp 256 || exit 1
  bench1 earth_det --no-f32 --scene earth --spp 256 --deterministic || exit 1;;
torchrun) step torchrun1 600 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-f32 > "$OUT/torchrun1.json" 2> "$OUT/torchrun1.err" || { tail -20 "$OUT/torchrun1.err"; exit 1; }; cat "$OUT/torchrun1.json";;
main1)  # the Python launcher under torchrun with one rank is a plain run (WORLD_SIZE 1: no exchange object); and the native one
  step main_py 300 python -u -m ray_tracing_amd --scene cornell_box --width 128 --height 128 --spp 32 --out "$OUT/main_py.png" > "$OUT/main_py.json" 2> "$OUT/main_py.err" || { tail -20 "$OUT/main_py.err"; exit 1; }; cat "$OUT/main_py.json"
  step main_native 300 ray-tracing_amd/rt_render --scene cornell_box --width 128 --height 128 --spp 32 --deterministic --out "$OUT/main_native.png" > "$OUT/main_native.json" 2> "$OUT/main_native.err" || { tail -20 "$OUT/main_native.err"; exit 1; }; cat "$OUT/main_native.json";;
esac
done
echo done | tee -a "$OUT/status.txt"
