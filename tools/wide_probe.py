"""Shard-0-of-N kernel times for the ordinary kernel and the cooperative kernel with G lanes per heavy pixel
(PT_WIDE_LOGG tuning knob): python tools/wide_probe.py [scene] [spp] [N,N,...]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
shards = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam)
torch.cuda.synchronize()
ref = {}
def run(name, flags):
    out = []
    for n in shards:
        best, fb = 1e30, None
        for _ in range(2):
            fb, ms = R.render(W, H, spp, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)
            best = min(best, ms)
        key = n
        if key not in ref: ref[key] = fb.clone()
        same = bool(torch.equal(fb.view(torch.int32), ref[key].view(torch.int32)))
        out.append(f"1/{n}: {best:7.1f} ms{'' if same else ' MISMATCH'}")
    print(f"{scene} {name:22s} " + "  ".join(out), flush=True)
run("ordinary (NO_COOP)", abi.PT_FLAG_NO_COOP)
run("coop, no split", abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT)
for lg in (1, 2, 3, 4):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    run(f"coop wide G={1 << lg}", abi.PT_FLAG_FORCE_COOP)
