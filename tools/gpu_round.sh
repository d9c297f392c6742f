#!/bin/bash
# One GPU-box session: parity tests -> bench -> rocprofv3 kernel trace -> PMC passes.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [tests|bench|prof|pmc ...]
# Every GPU step runs under its own time limit; a timeout / abort / segfault ends the script.
set -u
TAG=${1:-r01}; shift || true
WHAT=${*:-tests bench prof pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_SHORT="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32 ${BENCH_ARGS:-}"

fatal() { case $1 in 124|137|134|139|-6|-11) echo "step '$2' ended with $1: stopping" | tee -a "$OUT/status.txt"; exit "$1";; esac; }
step() {  # step <name> <seconds> <cmd...>
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a "$OUT/status.txt" >&2
  timeout -k 10 "$secs" "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/status.txt" >&2
  fatal $rc "$name"
  return $rc
}

cd "$REPO"
for w in $WHAT; do
case $w in
tests)
  step pytest_gpu 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > "$OUT/pytest_gpu.log" 2>&1 || { tail -30 "$OUT/pytest_gpu.log"; exit 1; }
  tail -3 "$OUT/pytest_gpu.log";;
bench)
  step bench 600 python -u bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
  cat "$OUT/bench.json";;
prof)
  cd /tmp
  step rocprof_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trace" -- python3 "$REPO"/$BENCH_SHORT > "$OUT/prof_trace.log" 2>&1 || tail -20 "$OUT/prof_trace.log"
  cd "$REPO";;
pmc)
  cd /tmp
  i=0
  for ctrs in \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
    "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
    "FETCH_SIZE" \
    "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
    i=$((i+1))
    step "pmc$i" 240 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc$i" -- python3 "$REPO"/$BENCH_SHORT > "$OUT/pmc$i.log" 2>&1 || tail -5 "$OUT/pmc$i.log"
  done
  cd "$REPO";;
esac
done
echo done | tee -a "$OUT/status.txt"
