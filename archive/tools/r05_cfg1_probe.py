"""config 1 (and two neighbours of it) with and without the cost probe, now that its samples are kept: kernel ms.   python tools/r05_cfg1_probe.py"""
import sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
packed, cam_args = scenes.build("smoke")
for W, H, spp in ((400, 225, 64), (800, 450, 64), (400, 225, 256)):
    cam = scenes.make_camera(cam_args, W, H)
    ds = R.DeviceScene(packed)
    R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
    for name, fl in (("default", 0), ("no probe", abi.PT_FLAG_NO_LPT)):
        ms = sorted(R.render(W, H, spp, ds, cam, flags=fl, timed=True)[1] for _ in range(8))
        print(f"smoke {W}x{H}x{spp} {name:9s}: min {ms[0]:6.2f} median {ms[4]:6.2f}", flush=True)
