#!/usr/bin/env python3
"""tests/golden/json_<name>.npz: the reference's JSON scene data codegen/data/<name>.json converted to the flat scene
form (flat.scene_from_json; our own serialisation, FlatScene.to_dict) + the oracle's outputs on it (64x64 primary
probe at time 0, xi 0.5; 32 samples of one lit pixel of a 32x32 frame).  The three files: scene_10 (25 spheres, 19
BVHNode records), scene_200_no_bvh (405 spheres in nested ObjectLists, no BVH), scene_500 (1005 spheres, 999 nodes).
`json_boxes` [n_nodes, 6] = the `bounding_box {min,max}` field every BVHNode record of the JSON carries (node order = the
flat scene's pre-order), data held by the reference: tests/test_json_boxes.py compares the boxes recomputed by the
flattener and by the oracle with them."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import flat
from oracle_lib import OracleScene
PIXEL = {"scene_10": (0, 11), "scene_200_no_bvh": None, "scene_500": None}
for name in PIXEL:
    fs = flat.scene_from_json(open(f"/root/reference/codegen/data/{name}.json").read())
    host = rt.FlatHostScene(fs)
    oc = OracleScene.from_desc(host.desc_ptr)
    ids, ts = oc.trace_primary(64, 64, 0.0, 0.5)
    px = PIXEL[name]
    if px is None:  # these scenes have background zero and no lights unless a sphere emits: take the brightest pixel of a small render
        acc, _ = oc.render(32, 32, 0, 8, threads=8)
        y, x = np.unravel_index(np.argmax(acc.sum(2)), acc.shape[:2])
        px = (int(x), int(y))
    smp = oc.sample_pixel(32, 32, px[0], px[1], 0, 32)
    out = "json_scene10" if name == "scene_10" else "json_" + name
    jb = np.array([fs.json_boxes[i] for i in range(len(fs.nodes))], dtype=np.float64).reshape(len(fs.nodes), 6)
    assert len(fs.json_boxes) == len(fs.nodes)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", out + ".npz"), flat_scene=json.dumps(fs.to_dict()), ids=ids, t=ts,
                        samples=smp, pixel=np.array(px), json_boxes=jb)
    print(name, "prims", fs.desc().n_prims, "hits", int((ids >= 0).sum()), "distinct", len(np.unique(ids)), "pixel", px, "mean sample", smp.mean(0))
