"""Experiment build (round 5): the chain priorities by ITERATIONS still needed (samples left x iterations per sample of this tile so far, in units of
spp; PT_PRIO_ITERS = onset16,f3,f2,f1) against the shipped rule (samples left).  Kernel ms.   python tools/r05_prio_iters.py lib"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 2:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("cornell", 1920, 1080, 1024, 1, 8), ("cornell", 1920, 1080, 1024, 2, 5), ("cornell", 1920, 1080, 1024, 4, 5), ("cornell", 1920, 1080, 256, 1, 6)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  {sys.argv[2]:16s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.2f}", flush=True)
else:
    here = Path(__file__).resolve().parent.parent / "path_tracer_amd"
    for rep in range(2):
        for name, env in (("shipped rule", {}), ("8,4,2,1", {"PT_PRIO_ITERS": "8,4,2,1"}), ("8,6,3,1.5", {"PT_PRIO_ITERS": "8,6,3,1.5"}), ("8,3,1.5,0.75", {"PT_PRIO_ITERS": "8,3,1.5,0.75"}),
                          ("4,4,2,1", {"PT_PRIO_ITERS": "4,4,2,1"}), ("0,4,2,1", {"PT_PRIO_ITERS": "0,4,2,1"}), ("0,6,3,1.5", {"PT_PRIO_ITERS": "0,6,3,1.5"}), ("4,8,4,2", {"PT_PRIO_ITERS": "4,8,4,2"})):
            subprocess.run([sys.executable, __file__, "x", name], env={**os.environ, **env, "PT_RENDER_LIB": str(here / sys.argv[1]), "PT_RENDER_LIB_ALLOW_OLDER": "1"})
