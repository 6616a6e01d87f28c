"""One wave's pace: an 8x8 frame (one tile = one wave) of the SmokeSphere scene aimed at the big glass ball — the pixels whose
sequential chains bound the large frames — at high spp; kernel ms per sample for the kernel variants the knobs select.
    python tools/lone_wave.py [spp]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
packed, cam_args = scenes.build("smoke")
cam_args = dict(cam_args, look_at=(0.0, 1.0, 0.0), vfov=1.0, aperture=0.0)
cam = scenes.make_camera(cam_args, 8, 8)
for name, env, flags in (("grid kernels (default: 4 lanes per wave)", {}, 0), ("grid kernels, the tile in ONE wave", {"PT_LANES_CAP": "0"}, 0), ("lists, ordinary kernels", {"PT_NO_GRID": "1"}, abi.PT_FLAG_NO_COOP),
                         ("lists, cooperative kernels", {"PT_NO_GRID": "1"}, abi.PT_FLAG_FORCE_COOP),
                         ("grid, scalar-cache kernels", {}, abi.PT_FLAG_NO_LDS)):
    for k, v in env.items():
        os.environ[k] = v
    ds = R.DeviceScene(packed)
    for k in env:
        del os.environ[k]
    R.render(8, 8, 16, ds, cam, flags=flags); torch.cuda.synchronize()
    ms = min(R.render(8, 8, spp, ds, cam, flags=flags | abi.PT_FLAG_NO_LPT, timed=True)[1] for _ in range(3))
    print(f"{name:42s} {ms:8.2f} ms for {spp} spp  = {ms / spp * 1e3:7.2f} us per sample of the slowest pixel", flush=True)
