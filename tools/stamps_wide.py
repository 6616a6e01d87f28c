"""s_memtime shares per wave-iteration (diagnostic build libpt_stamps.so) for the ordinary kernel and for the wide phase
(all tiles split, G lanes per pixel) on shard 0/N:  PT_RENDER_LIB=.../libpt_stamps.so python tools/stamps_wide.py cornell 256 64"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene, spp, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = abi.load_library()
lib.pt_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
W, H = 1920, 1080
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
def go(name, flags):
    lib.pt_debug_stamps(None, 1)
    fb, ms = R.render(W, H, spp, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)
    out = (C.c_ulonglong * 8)()
    lib.pt_debug_stamps(out, 0)
    prep, trav, shade, iters = out[0], out[1], out[2], max(out[3], 1)
    print(f"{scene} 1/{n} {name:18s} kernel {ms:7.1f} ms; wave-iterations {iters:.3e}; cycles/iteration: prepare {prep/iters:6.0f} traversal {trav/iters:6.0f} shade {shade/iters:6.0f} total {(prep+trav+shade)/iters:6.0f}", flush=True)
os.environ.pop("PT_SPLIT_TILES", None)
go("ordinary", abi.PT_FLAG_NO_COOP)
go("coop kernel G=1", abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT)
os.environ["PT_SPLIT_TILES"] = "-1"
for lg in (1, 2, 3):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    go(f"all wide G={1 << lg}", abi.PT_FLAG_FORCE_COOP)
