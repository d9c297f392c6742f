#!/usr/bin/env python3
"""tests/golden/json_scene10.npz: the reference's JSON scene data codegen/data/scene_10.json converted to the
flat scene form (flat.scene_from_json; our own serialisation, FlatScene.to_dict) + the oracle's outputs on it
(64x64 primary probe at time 0, xi 0.5; 32 samples of pixel (0,11) of a 32x32 frame)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import flat
from oracle_lib import OracleScene
fs = flat.scene_from_json(open("/root/reference/codegen/data/scene_10.json").read())
host = rt.FlatHostScene(fs)
oc = OracleScene.from_desc(host.desc_ptr)
ids, ts = oc.trace_primary(64, 64, 0.0, 0.5)
smp = oc.sample_pixel(32, 32, 0, 11, 0, 32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "json_scene10.npz"), flat_scene=json.dumps(fs.to_dict()), ids=ids, t=ts, samples=smp)
print("hits", int((ids >= 0).sum()), "distinct", len(np.unique(ids)), "mean sample", smp.mean(0))
