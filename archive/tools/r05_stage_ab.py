"""A/B of the two-stage probe (PT_PROBE_STAGE_A = samples of the unordered first stage; 0 = one stage): kernel ms.
    python tools/r05_stage_ab.py"""
import os
import subprocess
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 1:  # child: one setting (the knob is read once per process)
    import torch
    from path_tracer_amd import render as R, scenes
    for scene, W, H, spp, n in (("smoke", 1920, 1080, 1024, 1), ("smoke", 3840, 2160, 512, 8), ("smoke", 1920, 1080, 1024, 8), ("smoke", 1920, 1080, 512, 1)):
        packed, cam_args = scenes.build(scene)
        cam = scenes.make_camera(cam_args, W, H)
        ds = R.DeviceScene(packed)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
        ms = sorted(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(4))
        print(f"  stage_a={sys.argv[1]:>2s}  {scene} {W}x{H}x{spp} shard 0/{n}: {ms[0]:8.2f} (median {ms[1]:8.2f})", flush=True)
else:
    for rep in range(2):
        for a in ("0", "1", "2", "4"):
            subprocess.run([sys.executable, __file__, a], env={**os.environ, "PT_PROBE_STAGE_A": a})
