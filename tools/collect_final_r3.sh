#!/bin/bash
# After `gpurun -- 'bash tools/gpu_final_r3.sh && bash tools/gpu_prof_r3.sh'`: the closing numbers and profile summaries into profiles/r03_*
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r03_fin/bench.json profiles/r03_bench.json
{ cat $G/r03_fin/scenes.txt
  echo "--- other launches (Msamples/s f64, f32, launch ms)"
  for t in f64 f2k c128 c256 c512 s1k r1600; do echo "r03_fin_$t $(cat $G/r03_fin_$t/summary.txt | tail -1)"; done
  if [ -f $G/r03_others/summary.txt ]; then  # (tools/gpu_others_r3.sh)
    echo "--- scenes outside the BASELINE configs, 1024x1024 @ 64 spp per launch (Msamples/s f64, f32, ms per step, casts per sample)"
    cat $G/r03_others/summary.txt
  fi; } > profiles/r03_scenes.txt
cp $G/r03_fin_det/summary.txt profiles/r03_det_cost.txt
cp $G/r03_fin_cpu/cpu_configs.txt profiles/r03_cpu_configs.txt
bash tools/profiles_r3.sh
python - <<'PY'
import json
b = json.load(open("profiles/r03_bench.json"))
print("bench build", b["roofline"]["build"], "value", b["value"])
for n in ("cornell", "random", "final", "smoke", "earth", "perlin"):
    d = json.load(open(f"profiles/r03_{n}_summary.json"))
    print(n, {k: d.get(k) for k in ("build", "head")}, d.get("derived", {}).get("valu_lane_slot_pct"), d.get("derived", {}).get("hbm_bytes_per_launch"))
PY
