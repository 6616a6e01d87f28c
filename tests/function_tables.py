"""Function-level known-answer tables: one tiny scene per primitive x material so that a row of
the table is ONE call of hittable::hit() + material::emitted()/scatter() (render.hpp:58-89) on a recorded ray with a
recorded RNG state, and ONE call of camera::get_ray.  The inputs are generated here (deterministic); the expected outputs
are FROZEN in tests/golden/function_tables.npz by tests/golden/make_function_tables.py (oracle, portable math).  They
pin the oracle (and through it the device) against drift; they are not a pin to the reference itself."""
import numpy as np

from path_tracer_amd import abi
from path_tracer_amd.scene import (TextureAtlas, box, checker_texture, constant_medium, dielectric_material,
                                   image_texture, lambertian_material, lightsource_material, metal_material, pack,
                                   sphere, triangle, xy_rect, xz_rect, yz_rect)

N_RAYS = 256


def _image(atlas):
    y, x = np.mgrid[0:19, 0:31]
    rgb = np.stack([(x * 9 + y * 5) % 256, (x * 3 + 13 * y) % 256, (x * y + 7) % 256], axis=-1).astype(np.uint8)
    return image_texture.from_array(rgb, 3.0, atlas)


def cases():
    """name -> (PackedScene, (centre, extent) of the region the rays are aimed at)"""
    atlas = TextureAtlas()
    img = _image(atlas)
    checker = checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    out = {}
    out["sphere_lambertian_solid"] = ([sphere((0.1, 0.2, -1.0), 0.7, lambertian_material((0.7, 0.3, 0.3)))], ((0.1, 0.2, -1.0), 1.2))
    out["sphere_lambertian_image"] = ([sphere((0.1, 0.2, -1.0), 0.7, lambertian_material(img))], ((0.1, 0.2, -1.0), 1.2))
    out["sphere_moving_metal"] = ([sphere((0.0, 0.0, -1.0), (0.3, 0.5, -1.2), 0.1, 0.9, 0.6, metal_material((0.8, 0.6, 0.2), 0.35))], ((0.1, 0.2, -1.1), 1.3))
    out["sphere_dielectric"] = ([sphere((0.0, 0.0, 0.0), 1.0, dielectric_material(1.5, (1.0, 0.9, 0.8)))], ((0.0, 0.0, 0.0), 1.4))
    out["sphere_hollow_dielectric"] = ([sphere((0.0, 0.0, 0.0), -0.8, dielectric_material(1.3, (1, 1, 1)))], ((0.0, 0.0, 0.0), 1.2))
    out["xy_rect_lambertian_image"] = ([xy_rect(-1.0, 1.5, -0.5, 1.0, -2.0, lambertian_material(img))], ((0.25, 0.25, -2.0), 1.8))
    out["xz_rect_light"] = ([xz_rect(-1.0, 1.0, -2.0, 0.0, 2.5, lightsource_material((4, 3, 2)))], ((0.0, 2.5, -1.0), 1.6))
    out["yz_rect_lambertian_checker"] = ([yz_rect(-0.5, 1.5, -2.5, -0.5, -2.2, lambertian_material(checker))], ((-2.2, 0.5, -1.5), 1.6))
    out["triangle_lambertian"] = ([triangle((-0.5, 0.6, -1.2), (0.5, 0.6, -1.2), (0.0, 1.3, -0.9), lambertian_material((0.1, 0.2, 0.9)))], ((0.0, 0.85, -1.1), 0.9))
    out["triangle_badouel_lambertian"] = ([triangle((-0.5, 0.6, -1.2), (0.5, 0.6, -1.2), (0.0, 1.3, -0.9), lambertian_material((0.1, 0.2, 0.9)), "badouel")], ((0.0, 0.85, -1.1), 0.9))
    out["box_metal"] = ([box((1.2, -0.5, -2.5), (1.8, 0.7, -1.9), metal_material((0.7, 0.6, 0.5), 0.0))], ((1.5, 0.1, -2.2), 1.0))
    out["medium_sphere_isotropic"] = ([constant_medium(sphere((0.8, 0.9, -1.2), 0.6, lambertian_material((1, 1, 1))), 2.5, (0.9, 0.9, 1.0))], ((0.8, 0.9, -1.2), 1.0))
    out["medium_box_isotropic_checker"] = ([constant_medium(box((-1.9, -0.5, -0.9), (-1.3, 0.2, -0.3), lambertian_material((1, 1, 1))), 6.0, checker)], ((-1.6, -0.15, -0.6), 0.8))
    return {k: (pack(hs, atlas), region) for k, (hs, region) in out.items()}


def rays(name: str, region) -> np.ndarray:
    """N_RAYS recorded inputs (a ctypes array of PtBounceIn viewed through numpy bytes): aimed at the primitive, a third from
    inside its bounds, some grazing, some with very short / long direction vectors; fixed RNG states."""
    rng = np.random.default_rng(abs(hash(("function_tables", name))) % 2 ** 31 if False else sum(map(ord, name)) * 7919)
    centre, extent = np.float32(region[0]), np.float32(region[1])
    n = N_RAYS
    o = (rng.random((n, 3), dtype=np.float32) - 0.5) * 6 * extent + centre
    o[::3] = (rng.random((len(o[::3]), 3), dtype=np.float32) - 0.5) * 0.6 * extent + centre   # inside / very near
    target = (rng.random((n, 3), dtype=np.float32) - 0.5) * 1.6 * extent + centre
    d = (target - o).astype(np.float32)
    d[::7] *= np.float32(0.01)
    d[::11] *= np.float32(50.0)
    tm = rng.random(n, dtype=np.float32)
    st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    st[:3] = [1, 2463534242, 0xFFFFFFFF]
    att = rng.random((n, 3), dtype=np.float32)
    recs = (abi.PtBounceIn * n)()
    for k in range(n):
        recs[k].origin[:] = o[k].tolist()
        recs[k].dir[:] = d[k].tolist()
        recs[k].time = float(tm[k])
        recs[k].rng_state = int(st[k])
        recs[k].attenuation[:] = att[k].tolist()
    return recs


CAMERAS = {
    # name: (look_from, look_at, vup, vfov, aperture, focus_dist, t0, t1, width, height)   camera.hpp:67-87
    "main_cpp": ((13, 3, 3), (0, -1, 0), (0, 1, 0), 40.0, 0.04, None, 0.0, 1.0, 800, 480),
    "cornell": ((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 0.0, 800.0, 0.0, 1.0, 1920, 1080),
    "tilted_wide": ((0.3, 0.6, 2.5), (0, 0.2, -1), (0.2, 1, 0.1), 75.0, 0.3, 3.4, 0.25, 0.75, 333, 77),
}


def camera_inputs(name: str):
    w, h = CAMERAS[name][8:10]
    rng = np.random.default_rng(sum(map(ord, name)) * 104729)
    n = N_RAYS
    xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
    xy[:4] = [[0, 0], [w - 1, h - 1], [0, h - 1], [w - 1, 0]]
    st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    st[:4] = [0, 1, 0xFFFFFFFF, 2463534242]
    return xy, st
