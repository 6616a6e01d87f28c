"""cfg5's scene (100 k triangles through the triangle pool) against the oracle's brute-force scan on MANY sampled pixels: the 1080p frame at a low
sample count, N random pixels (mesh pixels preferred: the lower two thirds of the frame) re-rendered by the oracle.  A test tool, like tests/.
    python tools/check_cfg5_pixels.py [n_pixels] [spp] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from path_tracer_amd import abi, render as R, scenes
from oracle import binding as orc

n, spp, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 20000), (int(sys.argv[2]) if len(sys.argv) > 2 else 8), (int(sys.argv[3]) if len(sys.argv) > 3 else 7)
ps, cam = scenes.build("triangles", n_triangles=100_000)
w, h = 1920, 1080
c = scenes.make_camera(cam, w, h)
fb, ms = R.render(w, h, spp, R.DeviceScene(ps), c, timed=True)
fbn = fb.cpu().numpy()
rng = np.random.default_rng(seed)
xy = np.stack([rng.integers(0, w, n), rng.integers(0, 2 * h // 3, n)], axis=1).astype(np.int32)
orc.set_math(True)
t = time.time()
ref = orc.render_pixels(ps, c.c, w, h, spp, xy)
got = fbn[xy[:, 1], xy[:, 0]]
bad = int(np.any(got.view(np.uint32) != ref.view(np.uint32), axis=1).sum())
l = abi.load_library()
import ctypes as C
l.pt_build_id.restype = C.c_char_p
print(f"build {l.pt_build_id().decode()}: cfg5 scene 1920x1080x{spp} ({ms:.0f} ms on the GPU), {n} pixels = {n * spp} samples re-rendered by the oracle's brute-force scan in {time.time() - t:.0f} s: {bad} mismatching pixels")
sys.exit(1 if bad else 0)
