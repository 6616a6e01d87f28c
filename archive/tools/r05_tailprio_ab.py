"""Experiment build (round 5): every 64 iterations a wave sets its issue priority by what its slowest pixel still needs against what the wave has run
(PT_TAIL_PRIO=d3,d2,d1: needs more than 1/d3 of the iterations run so far -> 3, 1/d2 -> 2, 1/d1 -> 1).  Kernel ms of consecutive renders.   python tools/r05_tailprio_ab.py lib"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 2:
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n, reps in (("cornell", 1920, 1080, 1024, 1, 10), ("cornell", 1920, 1080, 1024, 4, 6), ("cornell", 1920, 1080, 1024, 8, 6), ("cornell", 1920, 1080, 256, 1, 8)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(reps)]
        print(f"  {sys.argv[2]:22s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.2f}  " + " ".join(f"{m:6.1f}" for m in ms), flush=True)
else:
    here = Path(__file__).resolve().parent.parent / "path_tracer_amd"
    for rep in range(2):
        E = lambda prio, onset, poll=64: {"PT_TAIL_PRIO": prio, "PT_TAIL_ONSET": str(onset), "PT_TAIL_POLL": str(poll)}
        for name, lib, env in (("shipped", "libpt_render.so", {}), ("2,4,8 from 8", sys.argv[1], E("2,4,8", 8)), ("2,4,8 from 6", sys.argv[1], E("2,4,8", 6)), ("2,4,8 from 7", sys.argv[1], E("2,4,8", 7)),
                               ("2,4,8 from 9", sys.argv[1], E("2,4,8", 9)), ("3,6,12 from 8", sys.argv[1], E("3,6,12", 8)), ("4,8,16 from 8", sys.argv[1], E("4,8,16", 8)),
                               ("2,4,16 from 8", sys.argv[1], E("2,4,16", 8)), ("2,8,32 from 8", sys.argv[1], E("2,8,32", 8)), ("2.5,5,10 from 8", sys.argv[1], E("2.5,5,10", 8)), ("2,4,8 from 8 poll 32", sys.argv[1], E("2,4,8", 8, 32))):
            subprocess.run([sys.executable, __file__, "x", name], env={**os.environ, **env, "PT_RENDER_LIB": str(here / lib), "PT_RENDER_LIB_ALLOW_OLDER": "1"})
