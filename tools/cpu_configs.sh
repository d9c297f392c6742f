#!/bin/bash
# CPU-side timings SURVEY §8(d) asks for beside the GPU numbers, on the host cores of the box it runs on:
#   C1 exactly as BASELINE states it (random_scene 800x450 @ 100 spp, all threads given to the job, at most 16), and
#   the single-thread per-path figure of benches/ray_tracing_benchmark.rs:6-23 (one camera path through
#   random_scene_light per iteration) as ns/path.
# Both through oracle/_build/oracle_render (the C++ f64 restatement; the Rust reference cannot be built here).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/${1:-cpu}; mkdir -p "$OUT"; cd "$REPO"
EXE=oracle/_build/oracle_render; IMG=ray-tracing_amd/assets/earthmap.ppm
T=${OMP_NUM_THREADS:-16}; [ "$T" -gt 16 ] && T=16
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), threads used: $T" | tee "$OUT/cpu_configs.txt"
timeout -k 10 600 $EXE random_scene 800 450 100 $T 0x5EED0001 $IMG | tail -1 | tee -a "$OUT/cpu_configs.txt"
timeout -k 10 300 $EXE random_scene_light 256 256 8 1 0x5EED0001 $IMG | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'bench':'ray tracing (one path, random_scene_light, 1 thread)','ns_per_path': round(1e3/d['msamples_per_s'],1), 'msamples_per_s': d['msamples_per_s']}))" | tee -a "$OUT/cpu_configs.txt"
