"""A/B of wave issue priorities by queue position (PT_PRIO_STEP, lane_acquire): kernel ms of shard 0 of N on one GPU.
    python tools/r05_prio.py [cornell|smoke] [width height spp] [steps ...]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, SPP = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
steps = [int(v) for v in sys.argv[5:]] or [0, 65536, 32768, 131072]
shards = [int(v) for v in os.environ.get("SHARDS", "1 4 8 16").split()]
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
ref = {}
for st in steps:
    os.environ["PT_PRIO_STEP"] = str(st)
    row = []
    for n in shards:
        best = 1e30
        for _ in range(3):
            fb, ms = R.render(W, H, SPP, ds, cam, shard_index=0, shard_count=n, timed=True)
            best = min(best, ms)
        row.append(best)
        if n == shards[-1]:
            key = fb.cpu().numpy().tobytes()
            ref.setdefault("fb", key)
            assert key == ref["fb"], "image changed"
    print(f"{scene} {W}x{H}x{SPP} prio_step {st:7d}: " + "  ".join(f"N={n}: {t:8.2f} ms" for n, t in zip(shards, row)), flush=True)
