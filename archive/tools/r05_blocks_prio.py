"""With the chain priorities in place: is four workgroups per CU still right for the headline family's launches with < 1.6 tiles per wave slot?
Kernel ms of shard 0 of N at PT_BLOCKS_PER_CU = (rule) / 4 / 8.   python tools/r05_blocks_prio.py"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
packed, cam_args = scenes.build("cornell")
W, H, spp = 1920, 1080, 1024
cam = scenes.make_camera(cam_args, W, H)
for n in (1, 2, 3, 4, 6, 8):
    out = []
    for b in ("", "4", "6", "8"):
        if b:
            os.environ["PT_BLOCKS_PER_CU"] = b
        else:
            os.environ.pop("PT_BLOCKS_PER_CU", None)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(5)]
        out.append(f"{b or 'rule'}: {sum(ms) / len(ms):7.2f}")
    print(f"cornell {W}x{H}x{spp} shard 0/{n}: workgroups per CU -> ms   " + "   ".join(out), flush=True)
