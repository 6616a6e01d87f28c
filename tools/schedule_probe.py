"""What the makespan model decides (tiles through the wide phase, lanes per pixel) and the resulting kernel time, for
config 1, the 1080p frame and its shards: python tools/schedule_probe.py [scene]"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
lib = abi.load_library()
packed, cam_args = scenes.build(scene)
ds = R.DeviceScene(packed)
def run(w, h, spp, n):
    cam = scenes.make_camera(cam_args, w, h)
    R.render(w, h, 16, ds, cam, shard_index=0, shard_count=n); torch.cuda.synchronize()
    ms = min(R.render(w, h, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
    out = (C.c_int32 * 2)()
    lib.pt_debug_schedule(ds.handle, out)
    tiles = ((w + 7) // 8) * ((h + 7) // 8) // n
    print(f"{scene} {w}x{h}x{spp} shard 0/{n}: {ms:8.1f} ms   wide tiles {out[0]:6d} of {tiles:6d}, G = {out[1]}", flush=True)
run(400, 225, 64, 1)
for n in (1, 2, 4, 8):
    run(1920, 1080, 256, n)
run(3840, 2160, 64, 1)
