"""Freezes the function-level known-answer tables (tests/function_tables.py) from the CPU oracle in portable-math mode:
    python tests/golden/make_function_tables.py   ->  tests/golden/function_tables.npz
Per primitive x material case: the recorded PtBounceIn rows and the PtBounceOut rows (hit flag, t, p, normal, front_face,
u, v, hittable, material, attenuation / emitted colour, scattered ray, RNG state after) as raw bytes; per camera: the 96
constructor-derived bytes (camera.hpp:67-87) and PtCameraRay rows of get_ray (camera.hpp:93-100)."""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import function_tables as FT  # noqa: E402
from oracle import binding as orc  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402

orc.build()
orc.set_math(True)
out = {}
for name, (ps, region) in FT.cases().items():
    recs = FT.rays(name, region)
    res = orc.bounce(ps, recs)
    out[f"bounce_in/{name}"] = np.frombuffer(bytes(recs), dtype=np.uint8)
    out[f"bounce_out/{name}"] = np.frombuffer(bytes(res), dtype=np.uint8)
    statuses = [res[k].status for k in range(len(recs))]
    print(f"{name:32s} statuses {sorted(set(statuses))}  hits {sum(s != 0 for s in statuses)}/{len(statuses)}")
for name, spec in FT.CAMERAS.items():
    look_from, look_at, vup, vfov, aperture, focus, t0, t1, w, h = spec
    if focus is None:
        d = np.float32(look_at) - np.float32(look_from)
        focus = float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
    cam = scenes.make_camera(dict(look_from=look_from, look_at=look_at, vup=vup, vfov=vfov, aperture=aperture,
                                  focus_dist=focus, time0=t0, time1=t1), w, h)
    xy, st = FT.camera_inputs(name)
    rays = orc.camera_rays(cam.c, w, h, xy, st)
    out[f"camera_fields/{name}"] = np.frombuffer(bytes(cam.c), dtype=np.uint8)
    out[f"camera_rays/{name}"] = np.frombuffer(bytes(rays), dtype=np.uint8)
np.savez_compressed(HERE / "function_tables.npz", **out)
print("wrote", HERE / "function_tables.npz", (HERE / "function_tables.npz").stat().st_size, "bytes")
