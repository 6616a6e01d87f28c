"""Chain speed-up and throughput cost of G lanes per pixel: all tiles forced through the split queue (PT_SPLIT_TILES=-1).
   1/64 shard = a few hundred lone waves (time ~ heaviest chain); whole frame at reduced spp = throughput."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
def t(flags, n, spp_):
    return min(R.render(W, H, spp_, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
os.environ.pop("PT_SPLIT_TILES", None)
print(f"{scene} ordinary            chain(1/64) {t(abi.PT_FLAG_NO_COOP, 64, spp):7.1f} ms   frame({spp//8} spp) {t(abi.PT_FLAG_NO_COOP, 1, spp//8):7.1f} ms", flush=True)
print(f"{scene} coop kernel, G=1    chain(1/64) {t(abi.PT_FLAG_FORCE_COOP|abi.PT_FLAG_NO_SPLIT, 64, spp):7.1f} ms   frame({spp//8} spp) {t(abi.PT_FLAG_FORCE_COOP|abi.PT_FLAG_NO_SPLIT, 1, spp//8):7.1f} ms", flush=True)
os.environ["PT_SPLIT_TILES"] = "-1"
for lg in (1, 2, 3, 4, 5):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    print(f"{scene} all tiles wide G={1<<lg:<2d} chain(1/64) {t(abi.PT_FLAG_FORCE_COOP, 64, spp):7.1f} ms   frame({spp//8} spp) {t(abi.PT_FLAG_FORCE_COOP, 1, spp//8):7.1f} ms", flush=True)
