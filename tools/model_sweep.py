"""Kernel time vs the two constants of the makespan model (PT_MODEL_FIXED, PT_MODEL_CHAIN) for several workloads.
   python tools/model_sweep.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
packed, cam_args = scenes.build("smoke")
lib = abi.load_library()
work = [(1920, 1080, 256, 1), (1920, 1080, 256, 8), (400, 225, 64, 1), (3840, 2160, 64, 1), (3840, 2160, 512, 8)]
def t(W, H, spp, n):
    d = R.DeviceScene(packed)
    cam = scenes.make_camera(cam_args, W, H)
    ms = min(R.render(W, H, spp, d, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
    out = (C.c_int32 * 2)()
    lib.pt_debug_schedule(d.handle, out)
    return ms, out[0], out[1]
R.render(400, 225, 16, R.DeviceScene(packed), scenes.make_camera(cam_args, 400, 225)); torch.cuda.synchronize()
print("workloads:", work, flush=True)
for fixed in (1200, 2400, 3600):
    for chain in (2400, 1000, 600):
        os.environ["PT_MODEL_FIXED"] = str(fixed); os.environ["PT_MODEL_CHAIN"] = str(chain)
        row = [t(*w) for w in work]
        print(f"fixed {fixed:4d} chain {chain:5d}: " + "  ".join(f"{ms:7.1f} ms ({tiles} tiles G={g})" for ms, tiles, g in row), flush=True)
