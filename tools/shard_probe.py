"""Times the render kernel for the whole frame and for shard 0 of N (what one GPU of an N-GPU job runs),
for both dequeue granularities — predicts multi-GPU scaling on one GPU.
  python tools/shard_probe.py [scene] [spp]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam)  # warm up
torch.cuda.synchronize()
for flags, name in ((abi.PT_FLAG_PIXEL_GRANULAR, "pixel-queue"), (abi.PT_FLAG_TILE_GRANULAR, "tile-queue")):
    base = None
    for n in (1, 2, 4, 8):
        ms = min(R.render(W, H, spp, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
        base = base or ms
        print(f"{scene} {name:12s} shard 0/{n}: {ms:8.2f} ms  -> {W*H*spp/n/ms/1e3:8.1f} Msamples/s per GPU, scaling eff {base/n/ms:5.2f}")
