#!/bin/bash
# Round-2 GPU session: new tests first (fail fast), then the whole GPU suite, bench (plain and under torchrun with 1 rank).
# Usage through gpurun:  bash tools/gpu_r2.sh <tag> [new|wip|all|bench|torchrun ...]
set -u
TAG=${1:-r02}; shift || true
WHAT=${*:-new wip all bench torchrun}
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
cd "$REPO"
PYT="python -u -m pytest -x -v --timeout 240 --timeout-method thread"
for w in $WHAT; do
case $w in
div) step pytest_div 300 $PYT tests/test_gpu_divide.py > "$OUT/pytest_div.log" 2>&1 || { tail -40 "$OUT/pytest_div.log"; exit 1; }; tail -3 "$OUT/pytest_div.log";;
scenes) step scenes 1100 bash tools/gpu_scenes.sh $TAG > "$OUT/scenes.txt" 2> "$OUT/scenes.err" || { tail -20 "$OUT/scenes.err"; exit 1; }; cat "$OUT/scenes.txt";;
benchq) step benchq 600 python -u bench.py --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/benchq.json" 2> "$OUT/benchq.err" || { tail -20 "$OUT/benchq.err"; exit 1; }; cat "$OUT/benchq.json";;
new) step pytest_new 600 $PYT tests/test_gpu_round2.py > "$OUT/pytest_new.log" 2>&1 || { tail -40 "$OUT/pytest_new.log"; exit 1; }; tail -3 "$OUT/pytest_new.log";;
wip) step pytest_wip 300 $PYT tests/test_zz_no_alloc.py > "$OUT/pytest_wip.log" 2>&1 || { tail -40 "$OUT/pytest_wip.log"; exit 1; }; tail -3 "$OUT/pytest_wip.log";;
all) step pytest_gpu 1000 $PYT tests -m gpu > "$OUT/pytest_gpu.log" 2>&1 || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }; tail -3 "$OUT/pytest_gpu.log";;
bench) step bench 600 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }; cat "$OUT/bench.json";;
torchrun) step torchrun1 600 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-f32 > "$OUT/torchrun1.json" 2> "$OUT/torchrun1.err" || { tail -20 "$OUT/torchrun1.err"; exit 1; }; cat "$OUT/torchrun1.json";;
esac
done
echo done | tee -a "$OUT/status.txt"
