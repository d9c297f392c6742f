#!/usr/bin/env python3
"""Build A/B variant libraries of librt_hip.so (different -D flags) into ray-tracing_amd/ab/ for
tools/gpu_ab.sh.  usage: python tools/build_ab.py name:DEF1,DEF2[:f64-unit-flag,flag] name2:DEF ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_amd as rt
from ray_tracing_amd import _build
for spec in sys.argv[1:]:
    name, _, rest = spec.partition(":")
    defs, _, flags = rest.partition(":")
    out = os.path.join(ROOT, "ray-tracing_amd", "ab", f"librt_hip_{name}.so")
    print(_build.build_hip(force=True, defines=[d for d in defs.split(",") if d], out=out, tag="_" + name,
                           f64_flags=[f for f in flags.split(",") if f]))
