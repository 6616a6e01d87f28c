"""Experiment build (round 5): the chain priorities in the grid kernels too (PT_PRIO_ALL=onset16,d3,d2,d1), on the launches that are bound by their
chains — shards of the 4K / 1080p SmokeSphere frame, config 1 — and on a whole frame.  Kernel ms.   python tools/r05_prio_grid.py lib"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 2:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("smoke", 3840, 2160, 512, 8, 5), ("smoke", 1920, 1080, 1024, 8, 5), ("smoke", 400, 225, 64, 1, 8), ("smoke", 3840, 2160, 512, 4, 4), ("smoke", 1920, 1080, 256, 1, 6)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  {sys.argv[2]:16s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.2f}  " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
else:
    here = Path(__file__).resolve().parent.parent / "path_tracer_amd"
    for rep in range(2):
        for name, env in (("off", {}), ("8,2,4,8", {"PT_PRIO_ALL": "8,2,4,8"}), ("0,2,4,8", {"PT_PRIO_ALL": "0,2,4,8"}), ("12,2,4,8", {"PT_PRIO_ALL": "12,2,4,8"}), ("16,4,8,16", {"PT_PRIO_ALL": "16,4,8,16"}), ("4,1.5,3,6", {"PT_PRIO_ALL": "4,1.5,3,6"})):
            subprocess.run([sys.executable, __file__, "x", name], env={**os.environ, **env, "PT_RENDER_LIB": str(here / sys.argv[1]), "PT_RENDER_LIB_ALLOW_OLDER": "1"})
