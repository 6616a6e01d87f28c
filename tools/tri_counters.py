"""Counters of the triangle pool from a diagnostic build (make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_TRI):
    PT_RENDER_LIB=path_tracer_amd/libpt_stamps.so python tools/tri_counters.py [spp] [width height]
per ray: grid tests, band (cheap) tests, exact tests from the band / always list; lanes busy per trip; fallbacks."""
import ctypes as C, os, sys
os.environ.setdefault('PT_TRICULL', '1')
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270)
lib = abi.load_library()
lib.pt_debug_tri.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
packed, cam_args = scenes.build("triangles", n_triangles=100_000)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
st = (C.c_int32 * 8)()
lib.pt_debug_tri_pool(C.byref(packed.desc), st)
print("pool: triangles", st[0], "always", st[1], "levels", list(st)[2:5], "cells per triangle", st[5] / 1000, "blob MB", st[6] * 16 / 1e6)
R.render(W, H, 1, ds, cam, flags=abi.PT_FLAG_NO_LPT); torch.cuda.synchronize()
lib.pt_debug_tri(None, 1)
fb, ms = R.render(W, H, spp, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)
o = (C.c_ulonglong * 12)()
lib.pt_debug_tri(o, 0)
scans, rays, grid, alw, lanes, b0, b1, b2 = [o[i] for i in range(8)]
rays = max(rays, 1)
print(f"{W}x{H}x{spp}: {ms:.1f} ms = {W*H*spp/ms/1e3:.2f} Msamples/s; wave scans {scans:.3e}, live rays per scan {rays/max(scans,1):.1f}")
print(f"per ray: grid survivors (exact tests) {o[8]/rays:.1f}; band: past the integer band test {o[10]/rays:.1f}, survivors (exact tests) {o[9]/rays:.1f}")
print(f"per ray: always list: pairs past the band test {o[7]/rays:.1f}, exact tests {o[11]/rays:.1f} (band trips of levels 1 and 2 are counted together)")
print(f"per ray: grid rounds {grid/rays:.1f}; band trips level 0 / levels 1 + 2: {b0/rays:.1f} / {b1/rays:.1f} (lanes busy per trip {lanes/max(b0+b1,1):.1f}); grid cells visited {alw/rays:.1f}")

if os.environ.get("PT_TRI_JSON"):  # the record bench.py prices the culled algorithm with (profiles/<tag>_tripool_counters.json)
    import json
    final = {"final": True, "round": int(os.environ["PT_FINAL_ROUND"])} if os.environ.get("PT_FINAL_ROUND") else {}
    lanes_per_trip = lanes / max(b0 + b1, 1)
    json.dump({**final, "scene": "triangles", "workload": f"{W}x{H}x{spp}",
               "note": "in-kernel counters of the triangle pool, diagnostic build (make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_TRI), tools/tri_counters.py",
               "per_ray": {"exact_tests": (o[8] + o[9] + o[11]) / rays, "grid_filter_tests": grid / rays * 64 * 4, "band_tests": (b0 + b1) / rays * lanes_per_trip * 4,
                           "always_tests": float(st[1]), "noise_radius_tests": (o[10] + o[7]) / rays, "grid_cells": alw / rays},
               "pool": {"triangles": st[0], "always": st[1], "levels": list(st)[2:5], "blob_bytes": st[6] * 16}},
              open(os.environ["PT_TRI_JSON"], "w"), indent=1)
