"""How fast is a short unordered launch (what the cost probe is)?  kernel ms of smoke / cornell at probe depth, whole tiles against single pixels.
    python tools/r05_probe_pace.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes

for scene, W, H, spp in (("smoke", 1920, 1080, 16), ("smoke", 400, 225, 4), ("cornell", 1920, 1080, 4), ("smoke", 1920, 1080, 64)):
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, W, H)
    ds = R.DeviceScene(packed)
    R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
    out = []
    for name, fl in (("no probe, whole tiles", abi.PT_FLAG_NO_LPT), ("no probe, single pixels", abi.PT_FLAG_NO_LPT | abi.PT_FLAG_PIXEL_GRANULAR), ("default", 0)):
        ms = min(R.render(W, H, spp, ds, cam, flags=fl, timed=True)[1] for _ in range(5))
        out.append(f"{name}: {ms:7.2f}")
    print(f"{scene} {W}x{H}x{spp}: " + "   ".join(out), flush=True)
