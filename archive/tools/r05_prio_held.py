"""Experiment build (round 5): a tile earns the chain priorities only after PT_PRIO_HELD iterations on it (light tiles taken late are done before that);
PT_PRIO_ONSET = k/16 of the queue.  Kernel ms.   python tools/r05_prio_held.py lib"""
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
        print(f"  {sys.argv[2]:18s} {scene} {W}x{H}x{spp} shard 0/{n}: mean {sum(ms) / len(ms):7.2f}", flush=True)
else:
    here = Path(__file__).resolve().parent.parent / "path_tracer_amd"
    for rep in range(2):
        for name, env in (("held 0", {}), ("held 256", {"PT_PRIO_HELD": "256"}), ("held 512", {"PT_PRIO_HELD": "512"}), ("held 1024", {"PT_PRIO_HELD": "1024"}), ("held 2048", {"PT_PRIO_HELD": "2048"}),
                          ("held 512 from 6", {"PT_PRIO_HELD": "512", "PT_PRIO_ONSET": "6"}), ("held 1024 from 4", {"PT_PRIO_HELD": "1024", "PT_PRIO_ONSET": "4"}), ("held 1024 from 0", {"PT_PRIO_HELD": "1024", "PT_PRIO_ONSET": "0"})):
            subprocess.run([sys.executable, __file__, "x", name], env={**os.environ, **env, "PT_RENDER_LIB": str(here / sys.argv[1]), "PT_RENDER_LIB_ALLOW_OLDER": "1"})
