"""One render of shard 0 of N (for rocprofv3 --pmc runs of the low-occupancy regime):
   python3 tools/shard_once.py scene spp N [flags]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import scenes
from path_tracer_amd import render as R
scene, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
fb, ms = R.render(W, H, spp, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)
print(f"{scene} shard 0/{n} {spp} spp: {ms:.2f} ms")
