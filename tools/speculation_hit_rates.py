"""How well could the sequential chain of a heavy pixel be SPECULATED across samples?  (CPU only: the oracle.)

A pixel's samples share one xorshift32 stream (render.hpp:95-101,130-133): sample i + 1 starts from M^d_i s_i, d_i = the draws of
sample i.  On the Cornell-style scene every scatter is lambertian (3 draws), so d_i = 5 + 3 (r_i - 1) with r_i the rays of the
sample (Appendix A of SURVEY.md): a lane group could run sample i + 1 from the states of the G - 1 likeliest r_i while sample i
is still being traced, and keep the one that guessed right.  This script measures the distribution of r_i on the heaviest
pixels of the 1080p x 1024 spp frame — the pixels that set the makespan floor of an 8-GPU job — and prints the
hit rate of the G - 1 likeliest guesses and the chain speed-up a depth-2 pipeline could reach with it.
    python tools/speculation_hit_rates.py [pixels sampled, default 1500]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import binding as orc  # noqa: E402  (a tool, like tests/: the checker, never the product)
from path_tracer_amd import scenes  # noqa: E402

W, H, SPP = 1920, 1080, 1024
n_probe = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ps, cam_args = scenes.build("cornell")
cam = scenes.make_camera(cam_args, W, H)
orc.set_math(True)
rng = np.random.default_rng(3)
xy = np.stack([rng.integers(0, W, n_probe), rng.integers(0, H, n_probe)], 1).astype(np.int32)
_, rays = orc.render_pixels_rays(ps, cam.c, W, H, SPP, xy)
order = np.argsort(rays)[::-1]
heavy = xy[order[:16]]
print(f"rays per pixel over {n_probe} sampled pixels: median {np.median(rays):.0f}, mean {rays.mean():.0f}, 99th pct {np.percentile(rays, 99):.0f}, max {rays.max()}")
# per-sample ray counts of the heavy pixels: the first k samples of a pixel are the same whatever SPP is, so cumulative differences
cum = np.zeros((SPP + 1, len(heavy)), dtype=np.int64)
for k in range(1, SPP + 1):
    cum[k] = orc.render_pixels_rays(ps, cam.c, W, H, k, heavy)[1]
r = np.diff(cum, axis=0).reshape(-1)
vals, counts = np.unique(r, return_counts=True)
p = counts / counts.sum()
top = np.argsort(p)[::-1]
print("rays per sample on the 16 heaviest pixels: mean %.2f;  P(r): " % r.mean() + ", ".join(f"r={vals[i]}: {p[i]:.3f}" for i in top[:10]))
for G in (2, 4, 8, 16, 32):
    hit = p[top[:G - 1]].sum()
    print(f"  G = {G:2d} lanes per pixel: the {G - 1:2d} likeliest draw counts cover {hit:.3f} of the samples -> depth-2 pipeline chain speed-up <= {1 / (1 - hit / 2):.2f}x "
          f"for {G}x the lane time")
