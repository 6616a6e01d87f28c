"""What each GPU of an N-GPU WEAK-scaling bench run does, on one GPU: shard 0/N of the sqrt(N)-scaled frame."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
packed, cam_args = scenes.build(scene)
ds = R.DeviceScene(packed)
for n in (1, 2, 4, 8):
    W, H = int(round(1920 * n ** 0.5)), int(round(1080 * n ** 0.5))
    cam = scenes.make_camera(cam_args, W, H)
    R.render(W, H, 4, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
    ms = min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
    print(f"{scene} N={n}: frame {W}x{H}, shard 0/{n}: {ms:7.1f} ms -> {W*H*spp/ms/1e3:9.1f} Msamples/s for the job if every rank takes as long", flush=True)
