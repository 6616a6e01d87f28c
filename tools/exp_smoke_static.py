"""Experiment: how much of the SmokeSphere scan is the moving spheres?  Renders the scene as is and with every
sphere made static (time0 == time1 == 0), same process, same box.   python tools/exp_smoke_static.py [spp]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1920, 1080
for variant in ("as is", "all static", "as is", "all static"):
    ps, cam = scenes.build("smoke")
    if variant == "all static":
        for i in range(ps.n_hittables):
            h = ps.hittables[i]
            if h.kind == abi.PT_HIT_SPHERE:
                h.f[3], h.f[4], h.f[5] = h.f[0], h.f[1], h.f[2]
                h.f[7] = h.f[8] = 0.0
    c = scenes.make_camera(cam, W, H)
    ds = R.DeviceScene(ps)
    R.render(W, H, spp, ds, c)
    fb, ms = R.render(W, H, spp, ds, c, timed=True)
    torch.cuda.synchronize()
    print(f"{variant:10s} {W}x{H}x{spp}: {ms:8.1f} ms = {W * H * spp / ms / 1e3:8.1f} Msamples/s", flush=True)
