"""Launch-to-launch spread of one frame in one process: kernel ms of N consecutive renders, under a few launcher settings.
    python tools/r05_jitter.py [scene] [W H spp] [n]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
for name, tun, flags in (("default", {}, 0), ("no probe (raster order)", {}, abi.PT_FLAG_NO_LPT), ("in-place walk", {"grid_walk": 1}, 0),
                         ("3 workgroups per CU", {"blocks_per_cu": 3}, 0), ("probe thrown away", {"probe_resume": -1}, 0)):
    ds = R.DeviceScene(packed, tuning=abi.tuning(**tun))
    R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
    ms = [R.render(W, H, spp, ds, cam, flags=flags, timed=True)[1] for _ in range(n)]
    ll = (abi.C.c_int32 * 4)()
    abi.load_library().pt_debug_last_launch(ds.handle, ll)
    print(f"{scene} {W}x{H}x{spp} {name:26s} launch {list(ll)}: " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
