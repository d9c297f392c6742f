"""Register / scratch budgets of the f64 render kernels, read from the built library's code-object notes (no GPU needed).

Why a test: the occupancy each variant runs at was chosen by measurement (ray-tracing_amd/csrc/rt_variants.hpp), and
two of the budgets are narrow.  The generic variant (final_scene) runs 4 waves/SIMD with 112 B of scratch per lane: the scratch of
the resident waves (4096 waves x 64 lanes x 112 B = 29.4 MB) just fits the 32 MB of L2, and at 128 B (33.5 MB) the memory-side
traffic of a launch doubles (8.7 -> 18.6 GB, measured in round 3 when a 16-byte callee frame was added by accident).  A change to the
device code that moves a budget shows here, before anyone profiles."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ray-tracing_amd", "librt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes():
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(LIB, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = {}
        for f in os.listdir(tmp):
            if not f.endswith("gfx950"):
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in txt.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k == "name":
                    cur = {"name": v}
                elif k in ("vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
                    cur[k] = int(v)
                elif k == "wavefront_size" and "name" in cur:
                    out[cur["name"]] = cur
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def find(notes, pattern):
    hits = [v for k, v in notes.items() if re.search(pattern, k)]
    assert len(hits) == 1, (pattern, [h["name"] for h in hits])
    return hits[0]


# feature masks of rt_variants.hpp: cornell 1400, smoke 1402, flat (u,v) spheres 6548, spheres + BVH 6549, noise 10256, generic 15871
WAVE = r"13render_kernelIdLj%dELb0ELb%dE"
SORTED = r"20render_kernel_sortedIdLj%dE"
BUDGETS = [  # (what, symbol pattern, most VGPRs (= 4 waves/SIMD), most scratch bytes per lane)
    ("cornell_box, block-sorted", SORTED % 1400, 128, 32),
    ("cornell_smoke, block-sorted", SORTED % 1402, 128, 32),
    ("cornell wave kernel (NO_SORT / DETERMINISTIC)", WAVE % (1400, 0), 128, 32),
    ("earth", WAVE % (6548, 0), 128, 48),
    ("random_scene", WAVE % (6549, 0), 128, 96),
    ("two_perlin_spheres", WAVE % (10256, 0), 128, 48),
    ("final_scene, 256-lane blocks", WAVE % (15871, 0), 128, 112),
    ("final_scene, one block per CU", WAVE % (15871, 1), 128, 112),
]


@pytest.mark.parametrize("what,pattern,vgprs,scratch", BUDGETS)
def test_f64_render_kernel_budgets(what, pattern, vgprs, scratch):
    k = find(kernel_notes(), pattern)
    assert k["vgpr_count"] <= vgprs, (what, k)
    assert k["private_segment_fixed_size"] <= scratch, (what, k, "scratch per lane grew: see this file's header")


def test_one_block_per_cu_fits_the_lds():
    k = find(kernel_notes(), WAVE % (15871, 1))
    assert k["group_segment_fixed_size"] <= 163840
