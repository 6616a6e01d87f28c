"""Latency of one loop iteration of a single wave on an idle GPU: render a one-tile frame (8x8) at high spp."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from path_tracer_amd import scenes, abi
from path_tracer_amd import render as R
from oracle import binding as orc
for scene in sys.argv[1:]:
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, 8, 8)
    ds = R.DeviceScene(packed)
    spp = 4096
    R.render(8, 8, 64, ds, cam); torch.cuda.synchronize()
    for flags in (abi.PT_FLAG_NO_COOP, 0):
        ms = min(R.render(8, 8, spp, ds, cam, flags=flags, timed=True)[1] for _ in range(3))
        orc.set_math(True)
        _, ctr = orc.render(packed, cam.c, 8, 8, 64, counters=True)
        d = ctr.as_dict()
        rays_per_sample = d["rays"] / d["samples"]
        it = spp * rays_per_sample  # mean lane iterations; the wave runs max over lanes (~1.1-1.3x)
        print(f"{scene} flags={flags}: {ms:.2f} ms for ~{it:.0f} mean lane-iterations -> <= {ms*1e3/it:.2f} us per iteration ({ms*1e3/it*2400:.0f} cycles)")
