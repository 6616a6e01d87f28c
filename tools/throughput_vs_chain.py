"""Same number of samples, different chain lengths: 1080p x S spp vs 4K x S/4 spp (4x the pixels, chains 4x shorter).
A large gap means the 1080p frame is bound by its heaviest pixels (makespan), not by throughput."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
packed, cam_args = scenes.build(scene)
ds = R.DeviceScene(packed)
for (W, H, s) in ((1920, 1080, spp), (3840, 2160, spp // 4), (7680, 4320, spp // 16)):
    cam = scenes.make_camera(cam_args, W, H)
    R.render(W, H, 4, ds, cam); torch.cuda.synchronize()
    for flags, name in ((0, "default"), (abi.PT_FLAG_NO_COOP, "ordinary kernel")):
        ms = min(R.render(W, H, s, ds, cam, flags=flags, timed=True)[1] for _ in range(2))
        print(f"{scene} {W}x{H} x {s:4d} spp {name:16s}: {ms:8.1f} ms  {W*H*s/ms/1e3:8.1f} Msamples/s", flush=True)
