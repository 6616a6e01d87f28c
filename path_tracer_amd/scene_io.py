"""Scene fixture format: the C-ABI tables of a PackedScene (+ optional camera arguments) in one .npz file, so the same
dump can feed the GPU path, the C++ facade and the oracle.  Arrays are stored with the exact
struct layouts of include/pt_render.h, so `np.load(f)["hittables"].tobytes()` is a PtHittable[]."""
from __future__ import annotations

import json

import numpy as np

from . import abi
from .scene import PackedScene, hittable_dtype, pack_tables

material_dtype = np.dtype([("kind", "<i4"), ("texture", "<i4"), ("color", "<f4", (3,)), ("param", "<f4"),
                           ("reserved", "<i4", (2,))])
texture_dtype = np.dtype([("kind", "<i4"), ("color0", "<f4", (3,)), ("color1", "<f4", (3,)), ("width", "<u4"),
                          ("height", "<u4"), ("offset", "<u4"), ("freq", "<f4"), ("reserved", "<i4")])
assert material_dtype.itemsize == 32 and texture_dtype.itemsize == 48


def save_scene(path: str, scene: PackedScene, camera_args: dict | None = None) -> None:
    h = np.frombuffer(bytes(scene.hittables), dtype=hittable_dtype)[:scene.n_hittables]
    m = np.frombuffer(bytes(scene.materials), dtype=material_dtype)[:scene.n_materials]
    t = np.frombuffer(bytes(scene.textures), dtype=texture_dtype)[:scene.n_textures]
    atlas = np.frombuffer(bytes(scene.atlas), dtype=np.uint8)[:scene.atlas_bytes]
    np.savez_compressed(path, abi_version=np.int32(abi.PT_ABI_VERSION), hittables=h, materials=m, textures=t, atlas=atlas,
                        camera=np.array(json.dumps(camera_args or {})))


def load_scene(path: str):
    """Returns (PackedScene, camera_args dict)."""
    z = np.load(path, allow_pickle=False)
    if int(z["abi_version"]) not in (1, abi.PT_ABI_VERSION):  # (version 2 changed PtTuning and added entry points: the tables are version 1's)
        raise ValueError(f"{path}: ABI version {int(z['abi_version'])} is not one this build reads (1, {abi.PT_ABI_VERSION})")
    mats = (abi.PtMaterial * max(1, len(z["materials"]))).from_buffer_copy(z["materials"].tobytes() or bytes(32))
    texs = (abi.PtTexture * max(1, len(z["textures"]))).from_buffer_copy(z["textures"].tobytes() or bytes(48))
    ps = pack_tables(z["hittables"], [mats[i] for i in range(len(z["materials"]))],
                     [texs[i] for i in range(len(z["textures"]))], z["atlas"].tobytes())
    return ps, json.loads(str(z["camera"]))
