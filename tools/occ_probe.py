"""Kernel ms of a shard vs resident workgroups per CU (PT_BLOCKS_PER_CU, read at scene creation).
   python tools/occ_probe.py scene W H spp shards"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
scene, W, H, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
R.render(W, H, 16, R.DeviceScene(packed), cam, shard_index=0, shard_count=n)
for b in (8, 5, 4, 3, 2, 1):
    os.environ["PT_BLOCKS_PER_CU"] = str(b)
    ds = R.DeviceScene(packed)
    ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2)]
    print(f"{scene} {W}x{H}x{spp} shard 0/{n}: <= {b} workgroups per CU: {min(ms):8.1f} ms", flush=True)
