import sys
sys.path.insert(0, '.')
import torch
from path_tracer_amd import abi, render as R, scenes
packed, cam_args = scenes.build("smoke")
ds = R.DeviceScene(packed)
for (W, H, spp) in ((3840, 2160, 512), (1920, 1080, 512)):
    cam = scenes.make_camera(cam_args, W, H)
    for n in (1, 2, 4, 8):
        row = []
        for fl in (0, abi.PT_FLAG_PIXEL_GRANULAR, abi.PT_FLAG_TILE_GRANULAR):
            R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n, flags=fl)
            row.append(min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, flags=fl, timed=True)[1] for _ in range(2)))
        print(f"{W}x{H}x{spp} shard 0/{n}: default {row[0]:8.1f}  pixel-granular {row[1]:8.1f}  tile-granular {row[2]:8.1f}", flush=True)
