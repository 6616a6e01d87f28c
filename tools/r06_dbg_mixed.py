"""round 6 debugging aid: repeated renders of the small scenes in every dequeue mode, each compared with the oracle (is the device deterministic?)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import scenes_small as S
from oracle import binding as orc
from path_tracer_amd import abi, scenes, render as R
orc.set_math(True)
names = sys.argv[1:] or ["mixed"]
for nm in names:
    ps, cam = S.ALL[nm]()
    c = scenes.make_camera(cam, 100, 60)
    ref = orc.render(ps, c.c, 100, 60, 12)
    ds = R.DeviceScene(ps)
    for name, fl in (("tile", abi.PT_FLAG_TILE_GRANULAR), ("pixel", abi.PT_FLAG_PIXEL_GRANULAR), ("pixel+stream", abi.PT_FLAG_PIXEL_GRANULAR | abi.PT_FLAG_FORCE_STREAM), ("stream", abi.PT_FLAG_FORCE_STREAM), ("default", 0)):
        nbad = 0
        for rep in range(6):
            a = R.render_host(100, 60, 12, ds, c, flags=fl)
            bad = np.argwhere((a.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
            nbad += len(bad) > 0
            if len(bad): print(nm, name, rep, "mismatching pixels:", [tuple(int(v) for v in b) for b in bad[:8]], flush=True)
        print(nm, name, "renders with mismatches:", nbad, "of 6", flush=True)
