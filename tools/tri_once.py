"""ONE render of the 100 k-triangle mesh through the triangle pool (the program to put behind `rocprofv3 --pmc ... --`):
    PT_TRICULL=1 python tools/tri_once.py [W H SPP]"""
import os, sys
os.environ.setdefault("PT_TRICULL", "1")
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
W, H, SPP = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270, 4)
packed, cam_args = scenes.build("triangles", n_triangles=100_000)
cam = scenes.make_camera(cam_args, W, H)
import time
_t0 = time.time()
ds = R.DeviceScene(packed)
torch.cuda.synchronize()
print(f"scene create {time.time() - _t0:.2f} s", flush=True)
fb, ms = R.render(W, H, SPP, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)
fb, ms = R.render(W, H, SPP, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)  # (second render: workspaces allocated, tables warm)
print(f"{W}x{H}x{SPP}: {ms:.1f} ms = {W * H * SPP / ms / 1e3:.3f} Msamples/s", flush=True)
