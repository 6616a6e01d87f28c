"""A/B of the sphere culling grid in ONE process (PT_NO_GRID is read when a scene is created):
    python tools/grid_ab.py [spp]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
packed, cam_args = scenes.build("smoke")
for W, H, n in ((1920, 1080, 1), (1920, 1080, 8), (3840, 2160, 1), (400, 225, 1)):
    cam = scenes.make_camera(cam_args, W, H)
    row = []
    for grid in (False, True, False, True):
        os.environ.pop("PT_NO_GRID", None)
        if not grid:
            os.environ["PT_NO_GRID"] = "1"
        ds = R.DeviceScene(packed)
        s = spp if W < 3000 else max(16, spp // 4)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n)
        ms = min(R.render(W, H, s, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
        row.append(f"{'grid' if grid else 'lists'} {ms:8.1f} ms ({W * H * s / n / ms / 1e3:7.1f} Msamples/s)")
    print(f"smoke {W}x{H} shard 0/{n}: " + " | ".join(row), flush=True)
