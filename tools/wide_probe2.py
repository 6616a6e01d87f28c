"""Shard 0/N times for G lanes per pixel with a FIXED fraction of the (cost-sorted) tiles split: PT_SPLIT_TILES knob."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
shards = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "4,8").split(",")]
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
def t(flags, n):
    return min(R.render(W, H, spp, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
for n in shards:
    tiles = (W // 8) * (H // 8) // n
    os.environ.pop("PT_SPLIT_TILES", None)
    print(f"{scene} 1/{n} ordinary {t(abi.PT_FLAG_NO_COOP, n):7.1f} ms   coop default {t(0, n):7.1f}", flush=True)
    for lg in (1, 2, 3):
        os.environ["PT_WIDE_LOGG"] = str(lg)
        row = []
        for frac in (0.05, 0.1, 0.2, 0.35, 0.6):
            os.environ["PT_SPLIT_TILES"] = str(int(tiles * frac))
            row.append(f"{int(frac*100):3d}%: {t(abi.PT_FLAG_FORCE_COOP, n):6.1f}")
        print(f"   G={1<<lg}  split " + "  ".join(row), flush=True)
