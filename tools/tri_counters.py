"""Counters of the triangle pool from a diagnostic build (make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_TRI):
    PT_RENDER_LIB=path_tracer_amd/libpt_stamps.so python tools/tri_counters.py [spp] [width height]
per ray: grid cells and exact tests, direction-map entries, survivors of the two band stages; rays per rho class."""
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
print("pool: triangles", st[0], "wide", st[1], "map entries (K)", list(st)[2:4], "map resolutions", (st[4] >> 20) & 1023, (st[4] >> 10) & 1023, st[4] & 1023, "cells per triangle", st[5] / 1000, "blob MB", st[6] * 16 / 1e6)
R.render(W, H, 1, ds, cam, flags=abi.PT_FLAG_NO_LPT); torch.cuda.synchronize()
lib.pt_debug_tri(None, 1)
fb, ms = R.render(W, H, spp, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)
o = (C.c_ulonglong * 16)()
lib.pt_debug_tri(o, 0)
# [0] scans [1] live rays [2] grid wave-steps [3] grid lane-steps [4] grid pairs (exact tests) [5] grid batches [6] map entries enumerated
# [7] past the integer band test [8] past the noise radius (exact tests) [9] rays through the second map [10] rays that streamed everything [11] band trips
scans, rays = max(o[0], 1), max(o[1], 1)
print(f"{W}x{H}x{spp}: {ms:.1f} ms = {W*H*spp/ms/1e3:.2f} Msamples/s; wave scans {scans:.3e}, live rays per scan {rays/scans:.1f}")
print(f"grid, per ray: cells visited {o[3]/rays:.1f}, exact tests {o[4]/rays:.1f}; per scan: wave-steps {o[2]/scans:.1f} (lanes per step {o[3]/max(o[2],1):.1f}), batches {o[5]/scans:.1f} (pairs per batch {o[4]/max(o[5],1):.1f})")
print(f"direction map, per ray: entries enumerated {o[6]/rays:.1f} in {o[11]/rays:.1f} trips, past the integer band test {o[7]/rays:.1f}, past the noise radius (exact tests) {o[8]/rays:.1f}")
print(f"rays through the second map {o[12]/rays:.4f}, through the third {o[9]/rays:.4f}, rays that streamed every record {o[10]/rays:.4f}")
# [13] camera rays served by their pixel's cached candidate list, [14] cached candidates they tested (round 6)
print(f"camera rays served from their pixel's cache: {o[13]/rays:.4f} of all rays, {o[14]/max(o[13],1):.1f} cached candidates (exact tests) each")

if os.environ.get("PT_TRI_JSON"):  # the record bench.py prices the culled algorithm with (profiles/<tag>_tripool_counters.json)
    import json
    final = {"final": True, "round": int(os.environ["PT_FINAL_ROUND"])} if os.environ.get("PT_FINAL_ROUND") else {}
    json.dump({**final, "scene": "triangles", "workload": f"{W}x{H}x{spp}",
               "note": "in-kernel counters of the triangle pool, diagnostic build (make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_TRI), tools/tri_counters.py",
               "per_ray": {"exact_tests": (o[4] + o[8] + o[14]) / rays, "cached_ray_share": o[13] / rays, "cached_exact_tests": o[14] / rays, "grid_filter_tests": 0.0, "band_tests": o[6] / rays,
                           "always_tests": 0.0, "noise_radius_tests": o[7] / rays, "grid_cells": o[3] / rays,
                           "grid_exact_tests": o[4] / rays, "band_exact_tests": o[8] / rays, "second_map_share": o[12] / rays, "third_map_share": o[9] / rays, "full_stream_share": o[10] / rays},
               "pool": {"triangles": st[0], "wide": st[1], "map_entries_k": list(st)[2:4], "map_res": [(st[4] >> 20) & 1023, (st[4] >> 10) & 1023, st[4] & 1023], "blob_bytes": st[6] * 16}},
              open(os.environ["PT_TRI_JSON"], "w"), indent=1)
