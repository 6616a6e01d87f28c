"""Frame-level soak of the scheduling paths (round 5): random (scene, frame size, samples, shard) — the frame rendered by the default path
(cost probe whose samples are kept, dilated cost map, chain priorities, heaviest-first order) must equal, bit for bit, the frame rendered with
none of it (no probe: PT_FLAG_NO_LPT, PtTuning.chain_priority = -1, probe_resume = -1).  Device against device: the parity tests compare with the oracle.
    python tools/soak_scheduling.py [n_cases] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes
import scenes_small as S

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
pool = {"cornell": scenes.build("cornell"), "smoke": scenes.build("smoke"), "field": S.sphere_field_scene(), "mixed": S.mixed_scene()}
plain = abi.tuning(chain_priority=-1, probe_resume=-1)
scene_cache = {}
bad = 0
pixels = 0
for case in range(n_cases):
    name = str(rng.choice(list(pool)))
    ps, cam_args = pool[name]
    W, H = int(rng.integers(64, 900)), int(rng.integers(64, 600))
    spp = int(rng.choice([16, 17, 24, 32, 48, 64, 96, 130]))
    n = int(rng.choice([1, 1, 1, 2, 3, 5]))
    idx = int(rng.integers(0, n))
    cam = scenes.make_camera(cam_args, W, H)
    if name not in scene_cache:
        scene_cache[name] = (R.DeviceScene(ps), R.DeviceScene(ps, tuning=plain))
    d0, d1 = scene_cache[name]
    a = R.render(W, H, spp, d0, cam, shard_index=idx, shard_count=n)
    b = R.render(W, H, spp, d1, cam, shard_index=idx, shard_count=n, flags=abi.PT_FLAG_NO_LPT)
    same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    pixels += a.numel() // 3
    if not same:
        bad += 1
        print(f"MISMATCH case {case}: {name} {W}x{H}x{spp} shard {idx}/{n}", flush=True)
    if case % 25 == 24:
        print(f"  {case + 1} cases, {pixels / 1e6:.1f} M pixels, {bad} mismatching frames", flush=True)
print(f"SCHEDULING SOAK: {n_cases} frames, {pixels / 1e6:.1f} M pixels compared, {bad} mismatching frames")
sys.exit(1 if bad else 0)
