"""With the probe's samples kept, how deep should the probe be?  PT_PROBE_SPP_MAX sweep on the grid-kernel workloads: kernel ms.
    python tools/r05_probe_depth.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes

for scene, W, H, spp, n in (("smoke", 1920, 1080, 1024, 1), ("smoke", 400, 225, 64, 1), ("smoke", 3840, 2160, 512, 8), ("smoke", 1920, 1080, 1024, 8)):
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, W, H)
    out = []
    for depth in (16, 32, 64, 128):
        os.environ["PT_PROBE_SPP_MAX"] = str(depth)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(5 if W * H * spp < 1e9 else 3))
        out.append(f"{depth}: {ms:8.2f}")
    print(f"{scene} {W}x{H}x{spp} shard 0/{n}: probe depth cap -> ms   " + "   ".join(out), flush=True)
