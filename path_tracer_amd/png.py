"""Dependency-free PNG writer for the output stage (reference: stbi_write_png, src/main.cpp:57).
zlib is in the Python standard library; no stb needed."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def write_png(path: str, rgb8: np.ndarray) -> None:
    """rgb8: [height][width][3] uint8, row 0 = top (what pt_tonemap_rgb8 produces)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, c = rgb8.shape
    if c != 3:
        raise ValueError("expected RGB")
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), rgb8.reshape(h, w * 3)], axis=1).tobytes()  # filter 0

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
