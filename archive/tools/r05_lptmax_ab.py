"""Headline family: tiles ordered by their ray count (default) or by their heaviest pixel (PT_LPT_MAX=1; with it the deeper probe and the
dilated cost map of the grid kernels) — now that the probe's samples are kept.  Kernel ms of consecutive renders.   python tools/r05_lptmax_ab.py"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("cornell", 1920, 1080, 1024, 1, 10), ("cornell", 1920, 1080, 256, 1, 10), ("cornell", 3840, 2160, 256, 1, 5), ("cornell", 1920, 1080, 1024, 4, 6)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  {sys.argv[1]:14s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.2f}  " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
else:
    for rep in range(2):
        for name, env in (("by ray count", {}), ("by max, 16", {"PT_LPT_MAX": "1"}), ("by max, 4", {"PT_LPT_MAX": "1", "PT_PROBE_SPP_MAX": "4"}), ("by max, 8", {"PT_LPT_MAX": "1", "PT_PROBE_SPP_MAX": "8"})):
            subprocess.run([sys.executable, __file__, name], env={**os.environ, **env})
