"""Decodes the two image textures BASELINE config 1 uses (/root/reference/src/main.cpp:133,145 load
images/Xilinx.jpg and images/SYCL.png) to RGB8 and stores the pixels as a data fixture.

Build container only (needs /root/reference and PIL); the GPU box and the tests read the .npz.
The reference decodes with stb_image forced to 3 channels (texture.hpp:104-105); stb is not in the image, so the
decoder here is PIL (libjpeg / zlib).  PNG decoding is lossless, so SYCL.png's texels are exactly the reference's;
a baseline JPEG's IDCT/upsampling may differ between decoders by a level or two per texel — which is why the DECODED
pixels are the fixture: oracle and GPU read the same bytes.

    python tests/golden/make_textures.py
"""
import hashlib
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference/images")
OUT = Path(__file__).resolve().parent / "cfg1_textures.npz"


def main() -> None:
    arrays, meta = {}, []
    for key, name in (("xilinx", "Xilinx.jpg"), ("sycl", "SYCL.png")):
        with Image.open(REF / name) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        arrays[key] = rgb
        meta.append(f"{name} {rgb.shape[1]}x{rgb.shape[0]} sha256(rgb8)={hashlib.sha256(rgb.tobytes()).hexdigest()[:16]}")
    arrays["meta"] = np.array("; ".join(meta) + f"; decoder PIL {Image.__version__}")
    np.savez_compressed(OUT, **arrays)
    print(OUT, OUT.stat().st_size, "bytes;", str(arrays["meta"]))


if __name__ == "__main__":
    main()
