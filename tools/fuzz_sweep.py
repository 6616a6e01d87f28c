"""One-off wider sweep of tests/test_gpu_fuzz.py's random scenes: python tools/fuzz_sweep.py first last"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from test_gpu_fuzz import random_scene
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
from oracle import binding as orc
first, last = int(sys.argv[1]), int(sys.argv[2])
orc.set_math(True)
bad = 0
FLAVOURS = (0, abi.PT_FLAG_FORCE_COOP, abi.PT_FLAG_NO_COOP, abi.PT_FLAG_NO_LDS, abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_NO_FASTDIV,
            abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_PIXEL_GRANULAR)
for seed in range(first, last):
    ps, cam = random_scene(seed, allow_image_on_triangle=(seed % 2 == 0))
    w, h, spp = 40 + seed % 9, 24 + seed % 5, 16 + seed % 7
    c = scenes.make_camera(cam, w, h)
    ref = orc.render(ps, c.c, w, h, spp)
    for f in FLAVOURS:
        got = R.render_host(w, h, spp, ps, c, flags=f)
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        if not same.all():
            bad += 1
            print(f"MISMATCH seed {seed} flags {f}: {int((~same).sum())} values", flush=True)
print(f"seeds {first}..{last - 1}: {bad} mismatching renders of {(last - first) * len(FLAVOURS)}")
# the wide phase (G lanes per pixel) needs >= 64 tiles and >= 16 spp for the probe pass that feeds it: larger frames,
# every tile forced through it (tuning knobs), three group sizes
import os
bad = n = 0
os.environ["PT_SPLIT_TILES"] = "-1"
for seed in range(first, last):
    ps, cam = random_scene(seed, allow_image_on_triangle=(seed % 4 == 0))
    w, h, spp = 72 + seed % 9, 64 + seed % 5, 16 + seed % 3
    c = scenes.make_camera(cam, w, h)
    ref = orc.render(ps, c.c, w, h, spp)
    for lg in (1, 3, 6):
        os.environ["PT_WIDE_LOGG"] = str(lg)
        got = R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_FORCE_COOP)
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        n += 1
        if not same.all():
            bad += 1
            print(f"MISMATCH (wide) seed {seed} G={1 << lg}: {int((~same).sum())} values", flush=True)
print(f"wide phase, seeds {first}..{last - 1}: {bad} mismatching renders of {n}")
