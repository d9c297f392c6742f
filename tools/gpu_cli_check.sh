#!/bin/bash
# End-to-end check of the launcher on the GPU box: one-go render == checkpointed render continued from its file.
set -eu
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/cli; mkdir -p "$OUT"; cd "$REPO"
A="--scene cornell_box --width 256 --height 256 --precision f64"
timeout -k 10 200 python -m ray_tracing_amd $A --spp 128 --out $OUT/a.png --save-accum $OUT/a.npy
timeout -k 10 200 python -m ray_tracing_amd $A --spp 64 --out $OUT/b64.png --checkpoint $OUT/b.rtck --pass-spp 32
timeout -k 10 200 python -m ray_tracing_amd $A --spp 128 --out $OUT/b.png --checkpoint $OUT/b.rtck --pass-spp 32 --resume --save-accum $OUT/b.npy
python - <<PY
import numpy as np
a=np.load("$OUT/a.npy"); b=np.load("$OUT/b.npy")
print("max rel diff", float(np.max(np.abs(a-b)/np.maximum(np.abs(a),1.0))), "mean", float(a.mean()/128))
assert np.allclose(a,b,rtol=1e-5,atol=1e-5)
pa=open("$OUT/a.png","rb").read(); pb=open("$OUT/b.png","rb").read()
print("png bytes", len(pa), len(pb), "identical" if pa==pb else "differ")
PY
