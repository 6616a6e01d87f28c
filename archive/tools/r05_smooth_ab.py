"""A/B of the dilated cost estimate (tile_dilate_kernel; PT_COST_DILATE = alpha, 0 = a tile's own estimate only): kernel ms of consecutive renders.
    python tools/r05_smooth_ab.py"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("smoke", 1920, 1080, 1024, 1, 20), ("smoke", 1920, 1080, 256, 1, 16), ("smoke", 3840, 2160, 256, 1, 8), ("smoke", 400, 225, 64, 1, 12), ("smoke", 3840, 2160, 1024, 1, 3)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  {sys.argv[1]:8s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.1f}  " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
else:
    for rep in range(2):
        for a, r in (("1.0", "1"), ("1.5", "1"), ("1.0", "2"), ("0.75", "1")):
            subprocess.run([sys.executable, __file__, a + "/r" + r], env={**os.environ, "PT_COST_DILATE": a, "PT_COST_RAD": r})
