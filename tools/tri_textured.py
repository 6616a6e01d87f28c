"""The 100 k-triangle mesh with an IMAGE texture on its ground sphere: the triangle-pool kernels that carry texture coordinates
(UV_WINNER).  Kernel ms of a 1920x1080 render:   python tools/tri_textured.py [spp]     (PT_RENDER_LIB selects the build)"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ps, cam_args = scenes.triangle_mesh_scene()
gi = ps.n_textures - 2                                   # the ground's grey texture (scenes.triangle_mesh_scene)
rng = np.random.default_rng(3)
atlas = rng.integers(0, 256, 32 * 32 * 3, dtype=np.uint8).tobytes()
t = ps.textures[gi]
t.kind, t.width, t.height, t.offset, t.freq = abi.PT_TEX_IMAGE, 32, 32, 0, 1.0
ps.atlas = (C.c_uint8 * len(atlas)).from_buffer_copy(atlas)
ps.atlas_bytes = len(atlas)
ps.desc = abi.PtSceneDesc(ps.hittables, ps.n_hittables, ps.materials, ps.n_materials, ps.textures, ps.n_textures, 0,
                          C.cast(ps.atlas, C.POINTER(C.c_uint8)), ps.atlas_bytes)
W, H = 1920, 1080
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(ps)
R.render(W, H, 2, ds, cam)
ms = [R.render(W, H, spp, ds, cam, timed=True)[1] for _ in range(3)]
print(f"textured ground, {W}x{H}x{spp}: {min(ms):.1f} ms = {W * H * spp / min(ms) / 1e3:.2f} Msamples/s  all {[round(m, 1) for m in ms]}", flush=True)
