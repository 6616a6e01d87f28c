"""Crops of the reference's OWN published output (/root/reference/doc/SmokeSphere.jpg, 800x480: the README's picture of the
main.cpp scene), stored as a data fixture for tests/test_reference_image_anchor.py.  Build container only.

The picture was rendered by an earlier revision of the reference (its checker squares are larger, its small spheres are laid
out by a different RNG sequence and none of them moves), so only regions whose content does not depend on those are kept:
sky, horizon, the logo sphere (image texture, cyclic frequency 5), the lettering on the big textured sphere, the top of the
mirror sphere (reflects sky), the upper monolith.  Pixels are data (decoded JPEG); no reference source is copied.

    python tests/golden/make_ref_doc_regions.py
"""
from pathlib import Path

import numpy as np
from PIL import Image

SRC = Path("/root/reference/doc/SmokeSphere.jpg")
OUT = Path(__file__).resolve().parent / "ref_doc_smokesphere_regions.npz"
# name: (x0, y0, x1, y1) in image coordinates (row 0 = top), 800x480
REGIONS = {"sky": (300, 10, 600, 50), "horizon_left": (0, 80, 150, 95), "sycl_sphere": (195, 12, 245, 68),
           "xilinx_text": (280, 190, 350, 225), "metal_sphere_top": (480, 100, 530, 125), "monolith_top": (645, 60, 690, 150)}

img = np.asarray(Image.open(SRC).convert("RGB"), dtype=np.uint8)
assert img.shape == (480, 800, 3)
out = {"size": np.array([800, 480])}
for name, (x0, y0, x1, y1) in REGIONS.items():
    out[f"box/{name}"] = np.array([x0, y0, x1, y1])
    out[f"rgb/{name}"] = img[y0:y1, x0:x1].copy()
np.savez_compressed(OUT, **out)
print(OUT, OUT.stat().st_size, "bytes")
