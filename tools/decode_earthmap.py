#!/usr/bin/env python3
"""Decode the reference's texture asset images/earthmap.jpg (baseline JPEG, 1024x512x3) to a raw
RGB8 binary PPM with Pillow, so neither the GPU box nor the C++ host needs a JPEG decoder.

The reference decodes it with image 0.23.7 -> jpeg-decoder 0.1.20 (src/texture.rs:68-74); decoders
may differ by +-1-2 LSB per texel (IDCT / chroma upsampling), which no reference test pins.

    python tools/decode_earthmap.py /root/reference/images/earthmap.jpg ray-tracing_amd/assets/earthmap.ppm
"""
import sys
from PIL import Image

src, dst = sys.argv[1], sys.argv[2]
img = Image.open(src).convert("RGB")
w, h = img.size
with open(dst, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (w, h))
    f.write(img.tobytes())
print(f"{src} -> {dst}: {w}x{h}, {w*h*3} bytes")
