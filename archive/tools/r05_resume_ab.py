"""A/B of the probe pass's samples kept (default) against thrown away (PT_NO_PROBE_RESUME=1; read when a scene is created): kernel ms.
    python tools/r05_resume_ab.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes

for scene, W, H, spp, n in (("cornell", 1920, 1080, 1024, 1), ("smoke", 1920, 1080, 1024, 1), ("smoke", 400, 225, 64, 1), ("cornell", 1920, 1080, 1024, 8), ("smoke", 3840, 2160, 512, 8)):
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, W, H)
    row = {}
    for rep in range(2):
        for mode in ("kept", "again"):
            if mode == "again":
                os.environ["PT_NO_PROBE_RESUME"] = "1"
            else:
                os.environ.pop("PT_NO_PROBE_RESUME", None)
            ds = R.DeviceScene(packed)
            R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
            ms = min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(5 if W * H * spp < 1e9 else 3))
            row.setdefault(mode, []).append(ms)
    print(f"{scene} {W}x{H}x{spp} shard 0/{n}: kept " + " ".join(f"{v:8.2f}" for v in row["kept"]) + "   rendered again " + " ".join(f"{v:8.2f}" for v in row["again"]), flush=True)
