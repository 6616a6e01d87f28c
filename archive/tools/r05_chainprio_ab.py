"""A/B of the headline family's issue priorities (longest remaining chain first; PT_NO_CHAIN_PRIO=1 = off, read when a scene is created): kernel ms.
    python tools/r05_chainprio_ab.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes

for scene, W, H, spp, n in (("cornell", 1920, 1080, 1024, 1), ("cornell", 1920, 1080, 256, 1), ("cornell", 3840, 2160, 256, 1), ("cornell", 1280, 720, 1024, 1), ("cornell", 1920, 1080, 1024, 2),
                            ("cornell", 1920, 1080, 1024, 4), ("cornell", 1920, 1080, 1024, 8), ("cornell", 400, 225, 64, 1), ("cornell", 1920, 1080, 64, 1)):
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, W, H)
    row = {}
    for rep in range(2):
        for mode in ("on", "off"):
            if mode == "off":
                os.environ["PT_NO_CHAIN_PRIO"] = "1"
            else:
                os.environ.pop("PT_NO_CHAIN_PRIO", None)
            ds = R.DeviceScene(packed)
            R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
            ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(6)]
            row.setdefault(mode, []).append(sum(ms) / len(ms))
    print(f"{scene} {W}x{H}x{spp} shard 0/{n}: priorities " + " ".join(f"{v:8.2f}" for v in row["on"]) + "   none " + " ".join(f"{v:8.2f}" for v in row["off"]) + f"   ({row['on'][0] / row['off'][0] - 1:+.1%})", flush=True)
