"""Follows paths through a fuzz scene with the CPU oracle and checks the device on every ray (tests/path_rays.py); prints
the rays that differ.  Needs a GPU and the oracle (a test tool, like tests/):
    python tools/bounce_hunt.py box_field|sphere_field|random SEED [rays] [generations]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from path_tracer_amd import abi, scenes
from oracle import binding as orc
import test_gpu_fuzz as F
from path_rays import follow_paths

kind, seed = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 12
make = {"box_field": F.random_box_field, "sphere_field": F.random_sphere_field, "random": lambda s: F.random_scene(s, False)}[kind]
ps, cam = make(seed)
c = scenes.make_camera(cam, 40, 24)
checked, bad = follow_paths(abi.load_library(), orc, ps, c.c, 40, 24, n, gens, seed, verbose=True)
for line in bad[:8]:
    print(line)
print(f"{checked} rays checked, {len(bad)} mismatches")
