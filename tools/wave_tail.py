"""The end of a frame launch, wave by wave (experiment build with per-wave stamps — gpurun_out/csrc_diag of round 5, not in the tree: every wave's
last tile (queue position), when it took it and when the wave ended; 100 MHz clock): how many waves are still running t ms before the end, and
what the last ones are working on.   PT_RENDER_LIB=path_tracer_amd/libpt_waves.so PT_RENDER_LIB_ALLOW_OLDER=1 python tools/wave_tail.py [scene W H spp]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
lib = abi.load_library()
lib.pt_debug_waves.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
for rep in range(3):
    lib.pt_debug_waves(None, 0, 1)
    out, ms = R.render(W, H, spp, ds, cam, timed=True)
    del out
    ll = (C.c_int32 * 4)()
    lib.pt_debug_last_launch(ds.handle, ll)
    nw = min(int(ll[0]) * 4, 32768)
    buf = (C.c_ulonglong * (4 * nw))()
    abi.check(lib.pt_debug_waves(buf, nw, 0), "pt_debug_waves")
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 4).astype(np.int64)
    ran = a[:, 3] > 0
    unit, t_take, t_end, n_tiles = a[ran, 0], a[ran, 1], a[ran, 2], a[ran, 3]
    t0 = t_take.min()
    end = (t_end - t0) / 1e5; take = (t_take - t0) / 1e5
    T = end.max()
    print(f"{scene} {W}x{H}x{spp}: {ms:.1f} ms; waves that ran {ran.sum()} of {nw}; tiles per wave {n_tiles.mean():.2f}; the launch's last wave ends at {T:.1f} ms")
    print("   waves still running at T - x ms: " + "  ".join(f"{x}: {(end > T - x).sum()}" for x in (1, 2, 5, 10, 15, 20, 25, 30, 40, 50)))
    last = np.argsort(-end)[:12]
    print("   the last waves (end ms | last tile's queue position of %d | taken at ms | ran ms): " % (unit.max() + 1)
          + "  ".join(f"{end[i]:.1f}|{unit[i]}|{take[i]:.1f}|{end[i] - take[i]:.1f}" for i in last))
    dur = end - take
    for lo, hi in ((0, 1000), (1000, 4000), (4000, 8000), (8000, 16000), (16000, 24000), (24000, 40000)):
        m = (unit >= lo) & (unit < hi)
        if m.sum():
            print(f"   last tiles at queue positions [{lo}, {hi}): {m.sum()} waves, taken at {take[m].mean():.1f} ms on average, ran {dur[m].mean():.1f} ms (max {dur[m].max():.1f})")
