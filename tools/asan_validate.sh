#!/bin/bash
# Build tools/asan_validate.cpp with the host library's sources and the CPU build of the device templates (tests/hostsim) under
# AddressSanitizer + UBSan, and run it:  bash tools/asan_validate.sh [cases] [seed]      (host code only; no GPU involved)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/rt_asan_validate
mkdir -p "$OUT"
H=$ROOT/ray-tracing_amd/host
g++ -std=c++17 -O1 -g -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=all -fno-omit-frame-pointer -ffp-contract=off -DRT_HOST_SIM -Wno-unknown-pragmas \
    -I"$ROOT/tests/hostsim" -I"$ROOT/ray-tracing_amd/csrc" -I"$ROOT/include" -I"$H" \
    "$ROOT/tools/asan_validate.cpp" "$ROOT/tests/hostsim/sim.cpp" "$H/scene_api.cpp" "$H/scenes.cpp" "$H/flatten.cpp" "$H/jpeg.cpp" "$H/capi.cpp" \
    -pthread -o "$OUT/asan_validate"
ASAN_OPTIONS=detect_leaks=0 "$OUT/asan_validate" "${1:-2000}" "${2:-1}" "$ROOT/ray-tracing_amd/assets/earthmap.ppm"
