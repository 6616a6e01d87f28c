"""Scene sources for the configs BASELINE.json names, built with the reference-shaped
constructors of `scene.py`.  Everything is deterministic (the reference's own xorshift32, binary32
arithmetic), so the same call gives the same tables here and on the GPU box.

  cornell_box()          cfg2  7 `box` + 1 `xy_rect`, diffuse light — oracle-expressible "Cornell-style"
  smoke_sphere_scene()   cfg1/cfg3/cfg4  the scene literal of /root/reference/src/main.cpp:67-161
  triangle_mesh_scene()  cfg5  N small triangles + emissive xy_rect + ground sphere
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from . import abi
from .scene import (TextureAtlas, box, camera, checker_texture, constant_medium, dielectric_material,
                    hittable_dtype, image_texture, lambertian_material, lightsource_material, metal_material,
                    pack, pack_tables, sphere, triangle, xy_rect)

f32 = np.float32
XORSHIFT_DEFAULT_SEED = 2463534242  # xorshift.hpp:18


class HostRNG:
    """LocalPseudoRNG (rtweekend.hpp:33-92) for host-side scene construction (main.cpp:76)."""

    def __init__(self, seed: int = XORSHIFT_DEFAULT_SEED):
        self.s = seed & 0xFFFFFFFF

    def next_u32(self) -> int:
        s = self.s
        s ^= s >> 7
        s ^= (s << 1) & 0xFFFFFFFF
        s ^= s >> 9
        self.s = s
        return s

    def float_t(self, lo=None, hi=None):
        x = f32(self.next_u32()) * f32(2.0 ** -32)  # uint32 -> float is round-to-nearest-even in numpy too
        if lo is None:
            return x
        return f32(lo) + (f32(hi) - f32(lo)) * x

    def vec_t(self, lo=None, hi=None) -> np.ndarray:
        v = np.array([self.float_t(), self.float_t(), self.float_t()], dtype=f32)
        if lo is None:
            return v
        return v * (f32(hi) - f32(lo)) + f32(lo)  # rtweekend.hpp:54-57


def cornell_box():
    """Cornell-style box in reference terms: axis-aligned `box`es + one `xy_rect`,
    sky background (always on, render.hpp:83-87), camera outside looking in through the open front.
    Returns (hittables, camera_args)."""
    white = lambertian_material((0.73, 0.73, 0.73))
    red = lambertian_material((0.65, 0.05, 0.05))
    green = lambertian_material((0.12, 0.45, 0.15))
    light = lightsource_material((15.0, 15.0, 15.0))
    hittables = [
        box((555, 0, 0), (556, 555, 555), green),        # left wall
        box((-1, 0, 0), (0, 555, 555), red),             # right wall
        box((213, 554, 227), (343, 554.5, 332), light),  # ceiling light
        box((0, -1, 0), (555, 0, 555), white),           # floor
        box((0, 555, 0), (555, 556, 555), white),        # ceiling
        xy_rect(0, 555, 0, 555, 555, white),             # back wall
        box((130, 0, 65), (295, 165, 230), white),       # short block
        box((265, 0, 295), (430, 330, 460), white),      # tall block
    ]
    cam = dict(look_from=(278, 278, -800), look_at=(278, 278, 0), vup=(0, 1, 0), vfov=40.0, aperture=0.0,
               focus_dist=800.0, time0=0.0, time1=1.0)
    return hittables, cam


CFG1_TEXTURES = Path(__file__).resolve().parent.parent / "tests" / "golden" / "cfg1_textures.npz"


def reference_textures():
    """Decoded RGB8 of the two images main.cpp:133,145 load (images/Xilinx.jpg 1024x512, images/SYCL.png 1280x559):
    a committed DATA fixture (tests/golden/make_textures.py decoded them once in the build container; the reference
    tree does not travel).  Returns (xilinx, sycl) uint8 [h][w][3]."""
    with np.load(CFG1_TEXTURES) as z:
        return z["xilinx"], z["sycl"]


def export_reference_textures(directory) -> list:
    """Writes the decoded reference images as binary PPM (Xilinx.ppm, SYCL.ppm) — the format the C++ host's
    image_texture::image_texture_factory reads (path_tracer_amd/include/pt/path_tracer.hpp)."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    out = []
    for name, rgb in zip(("Xilinx.ppm", "SYCL.ppm"), reference_textures()):
        path = directory / name
        path.write_bytes(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]) + np.ascontiguousarray(rgb).tobytes())
        out.append(path)
    return out


def _procedural_image(w: int, h: int, seed: int) -> np.ndarray:
    """Explicit stand-in for the two reference images (textures="procedural"): a deterministic gradient +
    block pattern, so image_texture::value is exercised with real row/column structure without any file."""
    y, x = np.mgrid[0:h, 0:w]
    r = (x * 255 // max(1, w - 1)).astype(np.uint8)
    g = (y * 255 // max(1, h - 1)).astype(np.uint8)
    b = ((((x // 8) + (y // 8) + seed) % 2) * 200 + 30).astype(np.uint8)
    return np.stack([r, g, b], axis=-1)


def smoke_sphere_scene(atlas: TextureAtlas | None = None, xilinx_rgb: np.ndarray | None = None,
                       sycl_rgb: np.ndarray | None = None, textures: str = "reference", arg_order: str = "rtl"):
    """The default scene of /root/reference/src/main.cpp:67-161 ("SmokeSphere").

    main.cpp:83 leaves the order of the two rng calls among a constructor's arguments unspecified (:87,:92 multiply
    two vec_t() draws elementwise: commutative, the order cannot show), so the reference's exact scene depends on
    its compiler.  Here the order is DEFINED and selectable: arg_order="rtl" (default, round 6) draws the z
    displacement before the x displacement, which is what g++ does — with it the oracle reproduces every printed
    digit of the counters SURVEY.md §3.3 recorded from the reference's own g++ 11.4 build
    (tests/test_survey_counters_anchor.py); "ltr" is rounds 1-5's scene (the same population, another draw).  The
    result has the population (≈490 small spheres: 40 % lambertian, 40 % moving lambertian,
    15 % metal, 5 % glass; pyramid; light; image-textured rect + sphere; glass, lambertian and metal
    big spheres; logo sphere; metal monolith; smoke ball).  The atlas follows texture.hpp:113-114,157: the {0,0,1}
    fallback texel, then Xilinx.jpg's texels at offset 1, then SYCL.png's at 1 + 1024*512.
    textures: "reference" = the decoded reference images (reference_textures()), "procedural" = small generated
    stand-ins; explicit xilinx_rgb / sycl_rgb arrays override either.  Returns (hittables, camera_args, atlas)."""
    if textures not in ("reference", "procedural"):
        raise ValueError("textures must be 'reference' or 'procedural'")
    if arg_order not in ("ltr", "rtl"):
        raise ValueError("arg_order must be 'ltr' or 'rtl'")
    if textures == "reference" and (xilinx_rgb is None or sycl_rgb is None):
        rx, rs = reference_textures()
        xilinx_rgb = rx if xilinx_rgb is None else xilinx_rgb
        sycl_rgb = rs if sycl_rgb is None else sycl_rgb
    atlas = atlas or TextureAtlas()
    hittables = []
    t = checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    hittables.append(sphere((0, -1000, 0), 1000, lambertian_material(t)))
    rng = HostRNG()
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.float_t()
            if arg_order == "rtl":  # main.cpp:83's constructor arguments evaluated last to first (what g++ does)
                cz = f32(b) + f32(0.9) * rng.float_t()
                cx = f32(a) + f32(0.9) * rng.float_t()
            else:
                cx = f32(a) + f32(0.9) * rng.float_t()
                cz = f32(b) + f32(0.9) * rng.float_t()
            center = np.array([cx, f32(0.2), cz], dtype=f32)
            d = center - np.array([4, 0.2, 0], dtype=f32)
            if np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > f32(0.9):
                if choose_mat < f32(0.4):
                    albedo = rng.vec_t() * rng.vec_t()
                    hittables.append(sphere(center, 0.2, lambertian_material(albedo)))
                elif choose_mat < f32(0.8):
                    albedo = rng.vec_t() * rng.vec_t()
                    center2 = center + np.array([0, rng.float_t(0, 0.25), 0], dtype=f32)
                    hittables.append(sphere(center, center2, 0.0, 1.0, 0.2, lambertian_material(albedo)))
                elif choose_mat < f32(0.95):
                    albedo = rng.vec_t(0.5, 1)
                    fuzz = rng.float_t(0, 0.5)
                    hittables.append(sphere(center, 0.2, metal_material(albedo, fuzz)))
                else:
                    hittables.append(sphere(center, 0.2, dielectric_material(1.5, (1.0, 1.0, 1.0))))
    # pyramid main.cpp:113-126
    hittables.append(triangle((6.5, 0.0, 1.30), (6.25, 0.50, 1.05), (6.5, 0.0, 0.80), lambertian_material((0.68, 0.50, 0.1))))
    hittables.append(triangle((6.0, 0.0, 1.30), (6.25, 0.50, 1.05), (6.5, 0.0, 1.30), lambertian_material((0.89, 0.73, 0.29))))
    hittables.append(triangle((6.5, 0.0, 0.80), (6.25, 0.50, 1.05), (6.0, 0.0, 0.80), lambertian_material((0.0, 0.0, 1))))
    hittables.append(triangle((6.0, 0.0, 0.80), (6.25, 0.50, 1.05), (6.0, 0.0, 1.30), lambertian_material((0.0, 0.0, 1))))
    # glowing ball main.cpp:129-130
    hittables.append(sphere((4, 1, 0), 0.2, lightsource_material((10, 0, 10))))
    # image-textured rect + sphere, then the three big spheres main.cpp:133-142
    xil = xilinx_rgb if xilinx_rgb is not None else _procedural_image(256, 128, 0)
    t = image_texture.from_array(xil, 1.0, atlas)
    hittables.append(xy_rect(2, 4, 0, 1, -1, lambertian_material(t)))
    hittables.append(sphere((4, 1, 2.25), 1, lambertian_material(t)))
    hittables.append(sphere((0, 1, 0), 1, dielectric_material(1.5, (1.0, 0.5, 0.5))))
    hittables.append(sphere((-4, 1, 0), 1, lambertian_material((0.4, 0.2, 0.1))))
    hittables.append(sphere((0, 1, -2.25), 1, metal_material((0.7, 0.6, 0.5), 0.0)))
    # logo sphere main.cpp:145-149
    syc = sycl_rgb if sycl_rgb is not None else _procedural_image(320, 140, 1)
    t = image_texture.from_array(syc, 5.0, atlas)
    hittables.append(sphere((-60, 3, 5), 4, lambertian_material(t)))
    # metallic monolith main.cpp:152-154
    hittables.append(box((6.5, 0, -1.5), (7.0, 3.0, -1.0), metal_material((0.7, 0.6, 0.5), 0.25)))
    # smoke ball main.cpp:157-161
    smoke_sphere = sphere((5, 1, 3.5), 1, lambertian_material((0.75, 0.75, 0.75)))
    hittables.append(constant_medium(smoke_sphere, 1, (1, 1, 1)))
    look_from, look_at = np.array([13, 3, 3], dtype=f32), np.array([0, -1, 0], dtype=f32)
    dd = look_at - look_from
    focus = float(np.sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]))  # main.cpp:179
    cam = dict(look_from=tuple(look_from), look_at=tuple(look_at), vup=(0, 1, 0), vfov=40.0, aperture=0.04,
               focus_dist=focus, time0=0.0, time1=1.0)
    return hittables, cam, atlas


def _xorshift_columns(n: int, k: int, seed: int) -> np.ndarray:
    """n independent xorshift32 streams (seeded by a Weyl sequence), k float_t() draws each."""
    s = ((np.arange(1, n + 1, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    s[s == 0] = 1
    out = np.empty((n, k), dtype=f32)
    for _ in range(8):  # decorrelate the low-entropy seeds
        s ^= s >> np.uint32(7); s ^= s << np.uint32(1); s ^= s >> np.uint32(9)
    for j in range(k):
        s ^= s >> np.uint32(7); s ^= s << np.uint32(1); s ^= s >> np.uint32(9)
        out[:, j] = s.astype(f32) * f32(2.0 ** -32)
    return out


def triangle_mesh_scene(n_triangles: int = 100_000, seed: int = 12345, n_colors: int = 64):
    """cfg5: n random small triangles (edge <= 0.3) in [-3,3]x[0,3]x[-3,3] with
    lambertian colours, a ground sphere and an emissive xy_rect.  Returns (PackedScene, camera_args)."""
    u = _xorshift_columns(n_triangles, 10, seed)
    v0 = np.stack([u[:, 0] * f32(6) - f32(3), u[:, 1] * f32(3), u[:, 2] * f32(6) - f32(3)], axis=1).astype(f32)
    v1 = (v0 + (u[:, 3:6] - f32(0.5)) * f32(0.3)).astype(f32)
    v2 = (v0 + (u[:, 6:9] - f32(0.5)) * f32(0.3)).astype(f32)
    color_id = np.minimum((u[:, 9] * f32(n_colors)).astype(np.int32), n_colors - 1)
    cols = _xorshift_columns(n_colors, 3, seed + 1)
    textures, materials = [], []
    for c in cols:
        t = abi.PtTexture(); t.kind = abi.PT_TEX_SOLID; t.color0[:] = [float(x) for x in c]
        m = abi.PtMaterial(); m.kind = abi.PT_MAT_LAMBERTIAN; m.texture = len(textures)
        textures.append(t); materials.append(m)
    t = abi.PtTexture(); t.kind = abi.PT_TEX_SOLID; t.color0[:] = [0.5, 0.5, 0.5]
    m = abi.PtMaterial(); m.kind = abi.PT_MAT_LAMBERTIAN; m.texture = len(textures)
    textures.append(t); materials.append(m)
    ground_mat = len(materials) - 1
    t = abi.PtTexture(); t.kind = abi.PT_TEX_SOLID; t.color0[:] = [8.0, 8.0, 8.0]
    m = abi.PtMaterial(); m.kind = abi.PT_MAT_LIGHTSOURCE; m.texture = len(textures)
    textures.append(t); materials.append(m)
    light_mat = len(materials) - 1
    h = np.zeros(n_triangles + 2, dtype=hittable_dtype)
    h["kind"][0] = abi.PT_HIT_SPHERE
    h["material"][0] = ground_mat
    h["f"][0, :9] = [0, -1000, 0, 0, -1000, 0, 1000, 0, 0]
    h["kind"][1:1 + n_triangles] = abi.PT_HIT_TRIANGLE
    h["material"][1:1 + n_triangles] = color_id
    h["f"][1:1 + n_triangles, 0:3] = v0
    h["f"][1:1 + n_triangles, 3:6] = v1
    h["f"][1:1 + n_triangles, 6:9] = v2
    h["kind"][-1] = abi.PT_HIT_XY_RECT
    h["material"][-1] = light_mat
    h["f"][-1, :5] = [-2, 2, 1, 3, -3.5]
    cam = dict(look_from=(0, 2.5, 9), look_at=(0, 1.2, 0), vup=(0, 1, 0), vfov=40.0, aperture=0.0, focus_dist=9.0,
               time0=0.0, time1=1.0)
    return pack_tables(h, materials, textures), cam


def make_camera(cam: dict, width: int, height: int) -> camera:
    """aspect = float(width)/height as main.cpp:181."""
    aspect = float(f32(width) / f32(height))
    return camera(cam["look_from"], cam["look_at"], cam["vup"], cam["vfov"], aspect, cam["aperture"],
                  cam["focus_dist"], cam["time0"], cam["time1"])


def build(name: str, **kw):
    """(PackedScene, camera_args) by config name: 'cornell', 'smoke', 'triangles'."""
    if name == "cornell":
        h, cam = cornell_box()
        return pack(h), cam
    if name == "smoke":
        h, cam, atlas = smoke_sphere_scene(**kw)
        return pack(h, atlas), cam
    if name == "triangles":
        return triangle_mesh_scene(**kw)
    raise KeyError(name)
