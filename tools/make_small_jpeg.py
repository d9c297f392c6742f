#!/usr/bin/env python3
"""tests/golden/earth_256x128.jpg: a small BASELINE JPEG (4:4:4, quality 90) for the GPU test that feeds an
ImageTexture through the in-repo decoder (ray-tracing_amd/host/jpeg.cpp).  Made from the committed raster
ray-tracing_amd/assets/earthmap.ppm (itself decoded from the reference's data file images/earthmap.jpg by
tools/decode_earthmap.py), so the reference tree is not needed to regenerate it."""
import os

from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = Image.open(os.path.join(ROOT, "ray-tracing_amd", "assets", "earthmap.ppm")).convert("RGB")
dst = os.path.join(ROOT, "tests", "golden", "earth_256x128.jpg")
src.resize((256, 128), Image.BILINEAR).save(dst, "JPEG", quality=90, subsampling=0, progressive=False, optimize=False)
print(dst, os.path.getsize(dst))
