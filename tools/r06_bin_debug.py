"""round 6, diagnostic build (make variant NAME=libpt_bindbg.so EXTRA=-DPT_BIN_DEBUG): per generation, the live rays whose wave scanned the pooled run in full."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["PT_RENDER_LIB"] = str(ROOT / "path_tracer_amd" / "libpt_bindbg.so")
os.environ["PT_RENDER_LIB_ALLOW_OLDER"] = "1"
os.environ.setdefault("PT_TRICULL", "1")
sys.path.insert(0, str(ROOT))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
W, H, SPP = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 8)
packed, cam_args = scenes.build("triangles", n_triangles=100_000)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
fb, ms = R.render(W, H, SPP, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)
lib = abi.load_library()
out = (C.c_uint32 * 4096)()
lib.pt_debug_bin_fallbacks(out)
print(f"{W}x{H}x{SPP}: {ms:.1f} ms")
for g in range(512):
    if out[2 * g]:
        print(f"gen {g}: {out[2 * g]} rays in {out[2 * g + 1]} waves without a request")
print("longest phase 2 of a wave per generation (us):", [round(out[1024 + g] / 100.0) for g in range(0, 330, 10)])
print("gen: live rays, requests in map 0 (dbg_base1 = 0: none), requests elsewhere")
for g in list(range(0, 40)) + list(range(40, 512, 10)):
    if out[2048 + g]:
        print(g, out[2048 + g], out[2560 + g], out[3072 + g])
