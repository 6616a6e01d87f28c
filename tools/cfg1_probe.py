"""BASELINE config 1 (default scene, 400x225, 64 spp): kernel time vs lanes per pixel in the wide phase (PT_WIDE_LOGG)
and vs a fixed split (PT_SPLIT_TILES)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
W, H, SPP = 400, 225, 64
packed, cam_args = scenes.build("smoke")
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
def t(flags=0):
    return min(R.render(W, H, SPP, ds, cam, flags=flags, timed=True)[1] for _ in range(3))
print(f"default {t():.1f} ms   ordinary kernel {t(abi.PT_FLAG_NO_COOP):.1f} ms   no split {t(abi.PT_FLAG_NO_SPLIT):.1f} ms", flush=True)
for lg in (2, 3, 4, 5, 6):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    os.environ.pop("PT_SPLIT_TILES", None)
    a = t()
    os.environ["PT_SPLIT_TILES"] = "-1"
    b = t()
    os.environ["PT_SPLIT_TILES"] = "300"
    c = t()
    print(f"G={1<<lg:2d}: model's split {a:6.1f} ms   all tiles wide {b:6.1f} ms   300 tiles wide {c:6.1f} ms", flush=True)
