"""Is the 496-hittable frame's slow mode (one launch in five takes 9 % longer) the clock?  A fixed ALU-bound torch op is timed before every
render: if it stretches with the slow renders the GPU's clock moved, if not the launch itself did.   python tools/r05_jitter2.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
packed, cam_args = scenes.build("smoke")
W, H, spp = 1920, 1080, 1024
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
x = torch.randn(1 << 20, device="cuda")
def proxy():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y = x
    e0.record()
    for _ in range(200):
        y = torch.sin(y) * 1.0001 + 0.1
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
proxy()
fb = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for i in range(24):
    p = proxy()
    out, ms = R.render(W, H, spp, ds, cam, timed=True)
    print(f"  proxy {p:7.3f} ms   render {ms:7.1f} ms   fb at {out.data_ptr():#x}", flush=True)
    del out
