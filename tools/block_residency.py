"""Where a frame launch's workgroups ran and when (diagnostic build: make -C path_tracer_amd/csrc variant NAME=libpt_blocks.so EXTRA=-DPT_STAMPS_BLOCKS):
per launch the kernel ms, how many workgroups started within 1 ms of the first, and the histogram of such workgroups per CU.
    PT_RENDER_LIB=path_tracer_amd/libpt_blocks.so PT_RENDER_LIB_ALLOW_OLDER=1 python tools/block_residency.py [scene W H spp n]"""
import ctypes as C
import sys
from collections import Counter
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
n = int(sys.argv[5]) if len(sys.argv) > 5 else 16
lib = abi.load_library()
lib.pt_debug_blocks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
for i in range(n):
    out, ms = R.render(W, H, spp, ds, cam, timed=True)
    del out
    ll = (C.c_int32 * 4)()
    lib.pt_debug_last_launch(ds.handle, ll)
    nb = min(int(ll[0]), 8192)
    buf = (C.c_ulonglong * (3 * nb))()
    abi.check(lib.pt_debug_blocks(buf, nb), "pt_debug_blocks")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 3)
    where, t0, t1 = a[:, 0], a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
    hw = (where & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (where >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7   # gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    early = (t0 - t0.min()) < 100_000            # 100 MHz: within 1 ms of the first workgroup
    per_cu = Counter(key[early].tolist())
    hist = Counter(per_cu.values())
    per_xcc = Counter(xcc[early].tolist())
    late = int((~early).sum())
    dur = (t1 - t0) / 1e5
    print(f"  {ms:7.1f} ms  workgroups {nb}, started late {late}; CUs seen {len(per_cu)}; workgroups per CU -> CUs {dict(sorted(hist.items()))}; per XCC {[per_xcc[k] for k in sorted(per_xcc)]}; "
          f"workgroup ms min {dur.min():.0f} / median {np.median(dur):.0f} / max {dur.max():.0f}", flush=True)
