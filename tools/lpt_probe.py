"""Tile order by the heaviest pixel (PT_LPT_MAX=1) vs by the sum (0), and the probe depth cap: kernel ms of a shard.
   python tools/lpt_probe.py scene W H spp shards"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
scene, W, H, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
kw = {"n_triangles": 100_000} if scene == "triangles" else {}
packed, cam_args = scenes.build(scene, **kw)
cam = scenes.make_camera(cam_args, W, H)
R.render(W, H, 16, R.DeviceScene(packed), cam, shard_index=0, shard_count=n)
for lmax, cap in ((0, 16), (1, 4), (1, 16), (1, 64), (0, 16)):
    os.environ["PT_LPT_MAX"] = str(lmax); os.environ["PT_PROBE_SPP_MAX"] = str(cap)
    ds = R.DeviceScene(packed)
    ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2)]
    print(f"{scene} {W}x{H}x{spp} shard 0/{n}: order by {'max pixel' if lmax else 'sum      '} probe cap {cap:3d}: {min(ms):8.1f} ms", flush=True)
