"""Soak run of the path-ray check (tests/path_rays.py) over many fuzz scenes: box fields, sphere fields, the random mixed
scenes with a slab pool forced into every stretch of rects / boxes, with and without image textures on triangles / media,
and (round 3) triangle fields through the triangle pool (PT_TRICULL=1: from 256 triangles).
Needs a GPU and the oracle (a test tool, like tests/).   python tools/soak_path_rays.py [scenes-per-kind-multiplier] [rays] [kinds, e.g. triangle]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from path_tracer_amd import abi, scenes
from oracle import binding as orc
import test_gpu_fuzz as F
from path_rays import follow_paths
lib = abi.load_library()
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
total = bad_total = 0
for kind, make, seeds, forced in (("triangle", F.random_triangle_field, range(9000, 9000 + 40 * mult), "tri"),
                                  ("box", F.random_box_field, range(5000, 5000 + 40 * mult), False),
                                  ("sphere", F.random_sphere_field, range(6000, 6000 + 30 * mult), False),
                                  ("random", lambda s: F.random_scene(s, False), range(7000, 7000 + 40 * mult), True),
                                  ("random-img", lambda s: F.random_scene(s, True), range(7100, 7100 + 20 * mult), True)):
    if only and kind not in only:
        continue
    for seed in seeds:
        ps, cam = make(seed)
        c = scenes.make_camera(cam, 40, 24)
        if forced == "tri": os.environ["PT_TRICULL"] = "1"
        elif forced: os.environ["PT_POOL_ALWAYS"] = "1"
        try:
            checked, bad = follow_paths(lib, orc, ps, c.c, 40, 24, n_rays, 12, seed)
        finally:
            os.environ.pop("PT_POOL_ALWAYS", None); os.environ.pop("PT_TRICULL", None)
        total += checked; bad_total += len(bad)
        if bad:
            print(kind, seed, len(bad), bad[0][:300], flush=True)
print(f"SOAK: {total} rays checked, {bad_total} mismatches", flush=True)
