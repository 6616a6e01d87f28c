"""Frame-level soak of the triangle pool's three renderers (round 6): random (triangle field or the 100 k mesh, frame size, samples, shard,
camera) — the frame of the persistent kernel WITH the camera rays' candidate cache (the default), of the same kernel WITHOUT it
(PtTuning.tri_cache = -1: every ray enumerates its direction-map list) and of the binned renderer (PtTuning.tri_binned = 1: generations,
direction-bin packets, the tail handed to persistent waves) must be the same bits.  Device against device — the parity tests compare all
three with the oracle at small sizes; this soak covers frame sizes at which pixels are narrow enough for the cache to be used.
    python tools/soak_tri_renderers.py [n_cases] [seed]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes
from test_gpu_fuzz import random_triangle_field

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
bad = pixels = 0
mesh = scenes.build("triangles", n_triangles=100_000)
for case in range(n_cases):
    if rng.random() < 0.25:
        name, (ps, cam_args) = "mesh", mesh
    else:
        seed = 8000 + int(rng.integers(0, 14))
        name, (ps, cam_args) = f"field {seed}", random_triangle_field(seed)
    W, H = int(rng.integers(96, 1400)), int(rng.integers(64, 800))
    spp = int(rng.choice([2, 3, 5, 8, 13]))
    n = int(rng.choice([1, 1, 2, 3]))
    idx = int(rng.integers(0, n))
    if rng.random() < 0.3:  # another view of the same scene: the cache is per pixel AND per camera
        cam_args = dict(cam_args, vfov=float(cam_args["vfov"]) * float(rng.uniform(0.3, 1.0)))
    cam = scenes.make_camera(cam_args, W, H)
    os.environ["PT_BIN_TAIL"] = str(rng.choice(["0", "0.25", "0.6"]))
    d = [R.DeviceScene(ps, tuning=abi.tuning(tri_min_run=256, **kw)) for kw in (dict(), dict(tri_cache=-1), dict(tri_binned=1))]
    fr = [R.render(W, H, spp, x, cam, shard_index=idx, shard_count=n) for x in d]
    torch.cuda.synchronize()
    same = all(torch.equal(fr[0].view(torch.int32), f.view(torch.int32)) for f in fr[1:])
    pixels += fr[0].numel() // 3
    if not same:
        bad += 1
        print(f"MISMATCH case {case}: {name} {W}x{H}x{spp} shard {idx}/{n}", flush=True)
    for x in d:
        x.close()
    if case % 10 == 9:
        print(f"  {case + 1} cases, {pixels / 1e6:.1f} M pixels, {bad} mismatching frames", flush=True)
print(f"TRIANGLE-RENDERER SOAK: {n_cases} frames x 3 renderers, {pixels / 1e6:.1f} M pixels compared, {bad} mismatching frames")
sys.exit(1 if bad else 0)
