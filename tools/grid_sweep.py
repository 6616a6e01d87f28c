"""Kernel ms of a frame for several sphere-grid cell sizes / margins (PtTuning.grid_cell, grid_margin).
    python tools/grid_sweep.py scene W H spp "cell values" "margin values" """
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
scene, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cells = [float(x) for x in sys.argv[5].split(",")]
margins = [float(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0.0]
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
print(f"{scene} {W}x{H}x{spp}: kernel ms by grid_cell (rows) and grid_margin (columns); 0 = the builder's default")
print("cell\\m " + "".join(f"{m:>9.2f}" for m in margins))
for c in cells:
    row = []
    for m in margins:
        kw = {}
        if c > 0: kw["grid_cell"] = c
        if m > 0: kw["grid_margin"] = m
        ds = R.DeviceScene(packed, tuning=abi.tuning(**kw) if kw else None)
        R.render(W, H, 16, ds, cam)
        row.append(min(R.render(W, H, spp, ds, cam, timed=True)[1] for _ in range(3)))
    print(f"{c:<7.2f}" + "".join(f"{x:9.1f}" for x in row), flush=True)
