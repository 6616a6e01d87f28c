"""Whole-frame kernel time vs a FIXED number of tiles through the wide phase and the group size (tuning knobs):
   python tools/split_sweep.py scene spp [N]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
def t():
    d = R.DeviceScene(packed)  # the tuning knobs are read when a scene is created
    return min(R.render(W, H, spp, d, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
tiles = (W // 8) * (H // 8) // n
for k in ("PT_SPLIT_TILES", "PT_WIDE_LOGG"): os.environ.pop(k, None)
print(f"{scene} {spp} spp shard 0/{n}: model {t():7.1f} ms", flush=True)
for lg in (2, 3, 4, 5):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    row = []
    for frac in [float(x) for x in os.environ.get("FRACS", "0,0.005,0.02,0.05,0.1,0.2,0.4").split(",")]:
        os.environ["PT_SPLIT_TILES"] = str(int(tiles * frac))
        row.append(f"{frac*100:5.2f}%:{t():6.1f}")
    print(f"  G={1<<lg:2d}  " + "  ".join(row), flush=True)
