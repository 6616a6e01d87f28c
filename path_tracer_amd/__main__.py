"""Command-line caller of the hot path — the role of the reference's src/main.cpp:61-197 (scene → render → PNG).

    python -m path_tracer_amd --scene cornell --width 800 --height 480 --spp 100 --out out.png
"""
import argparse
import time

from . import render as R
from . import scenes
from .png import write_png


def main() -> None:
    ap = argparse.ArgumentParser(prog="python -m path_tracer_amd")
    ap.add_argument("--scene", default="smoke", choices=["smoke", "cornell", "triangles"],
                    help="smoke = the default scene of the reference's main.cpp")
    ap.add_argument("--width", type=int, default=800)    # CMakeLists.txt:44-54 defaults
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--spp", type=int, default=100)      # main.cpp:186
    ap.add_argument("--depth", type=int, default=50)     # render.hpp:144
    ap.add_argument("--triangles", type=int, default=100_000)
    ap.add_argument("--out", default="out.png")          # main.cpp:57
    ap.add_argument("--textures", default="reference", choices=["reference", "procedural"],
                    help="smoke scene: the decoded reference images (tests/golden/cfg1_textures.npz) or generated stand-ins")
    ap.add_argument("--export-textures", metavar="DIR", help="write Xilinx.ppm / SYCL.ppm for the C++ host and exit (no GPU)")
    a = ap.parse_args()
    if a.export_textures:
        print(*scenes.export_reference_textures(a.export_textures), sep="\n")
        return
    import torch

    kw = {"n_triangles": a.triangles} if a.scene == "triangles" else {"textures": a.textures} if a.scene == "smoke" else {}
    packed, cam_args = scenes.build(a.scene, **kw)
    cam = scenes.make_camera(cam_args, a.width, a.height)
    t0 = time.perf_counter()
    fb, ms = R.render(a.width, a.height, a.spp, packed, cam, a.depth, timed=True)
    rgb8 = R.tonemap_rgb8(fb)
    torch.cuda.synchronize()
    write_png(a.out, rgb8.cpu().numpy())
    n = a.width * a.height * a.spp
    print(f"{a.scene}: {packed.n_hittables} hittables, {a.width}x{a.height}x{a.spp} spp -> {a.out}; "
          f"kernel {ms:.1f} ms = {n / ms / 1e3:.1f} Msamples/s (wall {time.perf_counter() - t0:.2f} s)")


if __name__ == "__main__":
    main()
