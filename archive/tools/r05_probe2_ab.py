"""A/B of the probe's second, ordered stage (PT_PROBE2_DIV: samples / div; 0 = off): kernel ms of consecutive renders.
    python tools/r05_probe2_ab.py"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("smoke", 1920, 1080, 1024, 1, 16), ("smoke", 3840, 2160, 512, 8, 8), ("smoke", 1920, 1080, 1024, 8, 8), ("smoke", 1920, 1080, 256, 1, 12)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  div={sys.argv[1]:>2s}  {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.1f}  " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
else:
    for a in ("0", "8", "16", "4"):
        subprocess.run([sys.executable, __file__, a], env={**os.environ, "PT_PROBE2_DIV": a})
