"""shard 0/N kernel time for a set of flag combinations: python tools/shard8.py scene spp N flags..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import scenes
from path_tracer_amd import render as R
scene, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
for f in sys.argv[4:]:
    ms = min(R.render(W, H, spp, ds, cam, flags=int(f), shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
    print(f"{scene} 1/{n} flags={int(f):4d}: {ms:8.2f} ms", flush=True)
