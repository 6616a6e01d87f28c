"""Where the grid kernels overtake the cooperative ones on the SmokeSphere scene: kernel ms of small frames and of shards with the
threshold at 0 (grid always) and at a value that never lets the grid in.   python tools/grid_min_tiles.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes

packed, cam_args = scenes.build("smoke")
cases = [(400, 225, 64, 1), (400, 225, 256, 1), (800, 450, 128, 1), (1920, 1080, 256, 8), (1920, 1080, 256, 4), (1920, 1080, 256, 2), (3840, 2160, 128, 8)]
for thr in ("0", "100000000"):
    os.environ["PT_GRID_MIN_TILES"] = thr
    ds = R.DeviceScene(packed)
    for (w, h, spp, n) in cases:
        cam = scenes.make_camera(cam_args, w, h)
        R.render(w, h, 8, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = min(R.render(w, h, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(3))
        tiles = ((w + 7) // 8) * ((h + 7) // 8) // n
        print(f"{'grid' if thr == '0' else 'coop + lists':13s} {w}x{h}x{spp} shard 0/{n} ({tiles} tiles): {ms:8.2f} ms", flush=True)
