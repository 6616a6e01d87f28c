"""Triangle-pool frames: the cost probe + cost-sorted order under the stratified deal of pixels (PT_LPT_SCATTER=1), now that the probe's samples
are kept (round 3 measured the probe as pure cost there).  Kernel ms.   python tools/r05_tri_lpt.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
packed, cam_args = scenes.build("triangles")
for W, H, spp, n in ((1920, 1080, 32, 1), (1920, 1080, 64, 1), (1920, 1080, 64, 8)):
    cam = scenes.make_camera(cam_args, W, H)
    out = []
    for mode in ("default", "probe"):
        if mode == "probe":
            os.environ["PT_LPT_SCATTER"] = "1"
        else:
            os.environ.pop("PT_LPT_SCATTER", None)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2)]
        out.append(f"{mode}: " + " ".join(f"{m:8.1f}" for m in ms))
    print(f"triangles {W}x{H}x{spp} shard 0/{n}: " + "   ".join(out), flush=True)
