"""Randomised scenes (every hittable kind, material and texture, random order so that kinds interleave in short runs,
media anywhere in the list, coincident surfaces) rendered by every kernel flavour and compared with the oracle
bit for bit."""
import contextlib
import os

import numpy as np
import pytest

from conftest import assert_bit_identical
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
from path_tracer_amd.scene import (TextureAtlas, box, checker_texture, constant_medium, dielectric_material,
                                   image_texture, lambertian_material, lightsource_material, metal_material, pack,
                                   sphere, triangle, xy_rect, xz_rect, yz_rect)

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def forced_pools():
    """PT_POOL_ALWAYS while a DeviceScene is created (pt_scene_create reads it): slab pools wherever two rects / boxes meet."""
    os.environ["PT_POOL_ALWAYS"] = "1"
    try:
        yield
    finally:
        del os.environ["PT_POOL_ALWAYS"]


@contextlib.contextmanager
def tri_pools(**knobs):
    """PT_TRICULL=1 (+ PT_TRI_* knobs) while scenes are created: the exact culling of long triangle runs, from 256 triangles on
    (the default threshold is 4096)."""
    env = {"PT_TRICULL": "1", **{k: str(v) for k, v in knobs.items()}}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def random_scene(seed: int, allow_image_on_triangle: bool):
    rng = np.random.default_rng(seed)
    seed = int(seed)
    atlas = TextureAtlas()
    img = image_texture.from_array(rng.integers(0, 256, (13, 17, 3), dtype=np.uint8), float(rng.choice([1.0, 2.5])), atlas)

    def color():
        return tuple(float(x) for x in rng.random(3))

    def texture():
        k = rng.integers(0, 3)
        return color() if k == 0 else checker_texture(color(), color()) if k == 1 else img

    def material(for_tri=False):
        k = rng.integers(0, 5)
        if k == 0:
            t = texture()
            if for_tri and not allow_image_on_triangle and t is img:
                t = color()
            return lambertian_material(t)
        if k == 1:
            return metal_material(color(), float(rng.random()))
        if k == 2:
            return dielectric_material(float(1.2 + rng.random()), color())
        if k == 3:
            return lightsource_material(tuple(float(4 * x) for x in rng.random(3)))
        return lambertian_material(color())

    def pt(scale=2.0):
        return tuple(float(x) for x in (rng.random(3) - 0.5) * 2 * scale)

    hs = [sphere((0, -100.5, 0), 100, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    for _ in range(int(rng.integers(6, 28))):
        k = rng.integers(0, 8)
        if k == 0:
            hs.append(sphere(pt(), float(0.15 + 0.4 * rng.random()), material()))
        elif k == 1:
            c = pt()
            hs.append(sphere(c, (c[0], c[1] + float(0.3 * rng.random()), c[2]), 0.0, 1.0, float(0.1 + 0.3 * rng.random()), material()))
        elif k == 2:
            a, b = sorted(rng.random(2) * 3 - 1.5), sorted(rng.random(2) * 3 - 1.5)
            cls = [xy_rect, xz_rect, yz_rect][rng.integers(0, 3)]
            hs.append(cls(float(a[0]), float(a[1]), float(b[0]), float(b[1]), float(rng.random() * 3 - 1.5), material()))
        elif k == 3:
            v0 = np.array(pt())
            hs.append(triangle(tuple(v0), tuple(v0 + (rng.random(3) - 0.5)), tuple(v0 + (rng.random(3) - 0.5)), material(True),
                               "badouel" if (seed % 5 == 3 and rng.random() < 0.5) else "moller_trumbore"))
        elif k == 4:
            p0 = np.array(pt(1.5))
            hs.append(box(tuple(p0), tuple(p0 + 0.1 + rng.random(3)), material()))
        elif k == 5:
            med_tex = color() if (rng.random() < 0.5 or not allow_image_on_triangle) else img
            if rng.random() < 0.5:
                hs.append(constant_medium(sphere(pt(1.5), float(0.3 + 0.5 * rng.random()), lambertian_material((1, 1, 1))),
                                          float(0.5 + 4 * rng.random()), med_tex))
            else:
                p0 = np.array(pt(1.5))
                hs.append(constant_medium(box(tuple(p0), tuple(p0 + 0.2 + rng.random(3)), lambertian_material((1, 1, 1))),
                                          float(0.5 + 4 * rng.random()), med_tex))
        elif k == 6 and len(hs) > 1:
            hs.append(hs[int(rng.integers(1, len(hs)))])  # an exact duplicate later in the list: equal-t ties
        else:
            hs.append(sphere(pt(), float(-(0.1 + 0.3 * rng.random())), dielectric_material(1.5, (1, 1, 1))))  # negative radius
    order = rng.permutation(len(hs) - 1) + 1
    hs = [hs[0]] + [hs[i] for i in order]
    cam = dict(look_from=(float(3 * rng.random() + 2), float(1 + rng.random()), float(3 * rng.random() + 2)), look_at=(0, 0, 0),
               vup=(0, 1, 0), vfov=float(35 + 30 * rng.random()), aperture=float(0.2 * rng.random() * (rng.random() < 0.5)),
               focus_dist=float(3 + 2 * rng.random()), time0=0.0, time1=float(rng.random() < 0.7))
    return pack(hs, atlas), cam


@pytest.mark.parametrize("seed", range(10))
def test_random_scene_every_kernel_flavour(orc, seed):
    ps, cam = random_scene(1000 + seed, allow_image_on_triangle=(seed % 2 == 0))
    w, h, spp = 45, 27, 20
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    for name, flags in (("default", 0), ("cooperative", abi.PT_FLAG_FORCE_COOP), ("no-coop", abi.PT_FLAG_NO_COOP),
                        ("scalar", abi.PT_FLAG_NO_LDS), ("stream", abi.PT_FLAG_FORCE_STREAM),
                        ("stream+tile", abi.PT_FLAG_FORCE_STREAM | abi.PT_FLAG_TILE_GRANULAR),
                        ("plain-div", abi.PT_FLAG_NO_FASTDIV), ("coop+pixel", abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_PIXEL_GRANULAR)):
        assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"seed {seed} {name}")
    # the same scene with a slab pool for every stretch of rects / boxes, however short (the cost model would not build them
    # here): pools between sphere runs, triangles and media, ties against duplicates across kinds
    with forced_pools():
        ds = R.DeviceScene(ps)
    for name, flags in (("pools, LDS", abi.PT_FLAG_NO_COOP), ("pools, scalar cache", abi.PT_FLAG_NO_LDS), ("pools, default", 0)):
        assert_bit_identical(R.render_host(w, h, spp, ds, c, flags=flags), ref, f"seed {seed} {name}")


@pytest.mark.parametrize("seed", range(4))
def test_random_scene_sharded(orc, seed):
    from dist_util import unshard_reference
    ps, cam = random_scene(2000 + seed, allow_image_on_triangle=False)
    w, h, spp = 50, 30, 16
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    parts = [R.render_host(w, h, spp, ps, c, shard_index=i, shard_count=3) for i in range(3)]
    assert_bit_identical(unshard_reference(np.stack(parts), w, h, 3), ref, f"seed {seed} 3 shards")


def random_sphere_field(seed: int):
    """Runs of small spheres big enough for the culling grid (>= 48), with everything the grid has to get right varied at
    random: radii spread (so the "small" threshold and the margin differ), flat or tall fields, moving fractions, touching and
    duplicated spheres, negative radii, big spheres inside the field, large coordinates, a second sphere run, other kinds
    before / between / after, and cameras inside, near, far and very far (beyond the grid's rlimit: full-list fallback)."""
    rng = np.random.default_rng(seed)

    def color():
        return tuple(float(x) for x in rng.random(3))

    def material():
        k = rng.integers(0, 4)
        return (lambertian_material(color()) if k == 0 else metal_material(color(), float(0.4 * rng.random())) if k == 1
                else dielectric_material(1.5, (1, 1, 1)) if k == 2 else lambertian_material(checker_texture(color(), color())))

    scale = float(10.0 ** rng.integers(-1, 3))          # field size 0.1 .. 100
    centre = np.array([(rng.random() - 0.5) * 50 * (seed % 3 == 0) for _ in range(3)])
    r_med = 0.02 * scale * (0.5 + rng.random())
    tall = rng.random() < 0.4
    n = int(rng.integers(60, 400))
    hs = [sphere(tuple(centre + [0, -1000 * scale, 0]), 1000 * scale - 0.01 * scale, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    if rng.random() < 0.5:
        hs.append(xz_rect(float(centre[0] - scale), float(centre[0] + scale), float(centre[2] - scale), float(centre[2] + scale),
                          float(centre[1] + 1.5 * scale), lightsource_material((3, 3, 3))))
    first_run = len(hs)
    for i in range(n):
        c = centre + np.array([(rng.random() - 0.5) * scale, rng.random() * scale * (1.0 if tall else 0.08), (rng.random() - 0.5) * scale])
        r = float(r_med * (0.4 + 1.4 * rng.random()))
        if rng.random() < 0.03:
            r *= 6                                            # larger than 4 x the median: stays in the "big" list
        if rng.random() < 0.04:
            r = -r                                            # negative radius
        if rng.random() < 0.35:
            c1 = c + (rng.random(3) - 0.5) * 3 * r_med
            hs.append(sphere(tuple(c), tuple(c1), 0.0, 1.0, r, material()))
        else:
            hs.append(sphere(tuple(c), r, material()))
        if rng.random() < 0.03:
            hs.append(hs[int(rng.integers(first_run, len(hs)))])  # duplicate: equal-t tie, the earlier one must win
    if rng.random() < 0.6:                                      # interrupt the run, then a second (short or long) sphere run
        p0 = centre + (rng.random(3) - 0.5) * scale * 0.5
        hs.append(box(tuple(p0), tuple(p0 + 0.1 * scale), material()))
        for i in range(int(rng.choice([3, 70]))):
            c = centre + np.array([(rng.random() - 0.5) * scale, rng.random() * scale * 0.3, (rng.random() - 0.5) * scale])
            hs.append(sphere(tuple(c), float(r_med * (0.5 + rng.random())), material()))
    if rng.random() < 0.5:
        hs.append(constant_medium(sphere(tuple(centre + [0, 0.2 * scale, 0]), float(0.2 * scale), lambertian_material((1, 1, 1))),
                                  float(2.0 / scale), color()))
    dist = float(rng.choice([0.3, 1.5, 40.0, 4000.0])) * scale
    frm = centre + np.array([dist * 0.7, dist * 0.35 + 0.05 * scale, dist * 0.6])
    cam = dict(look_from=tuple(float(x) for x in frm), look_at=tuple(float(x) for x in centre + [0, 0.05 * scale, 0]), vup=(0, 1, 0),
               vfov=float(min(70.0, 2 * np.degrees(np.arctan(0.7 * scale / max(dist, 1e-6))) + 5.0)),
               aperture=float(0.02 * scale * (rng.random() < 0.3)), focus_dist=float(max(dist, 0.1 * scale)),
               time0=float(-0.3 * (seed % 7 == 5)), time1=1.0)
    return pack(hs), cam


@pytest.mark.parametrize("walk", [1, 2])
@pytest.mark.parametrize("seed", range(12))
def test_random_sphere_fields_through_the_culling_grid(orc, lib, seed, walk, monkeypatch):
    monkeypatch.setenv("PT_GRID_WALK", str(walk))  # both sphere-grid walks (PtTuning.grid_walk): in place, and through the pair queue
    ps, cam = random_sphere_field(3000 + seed)
    w, h, spp = 40, 24, 8
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    # frames this small keep the cooperative kernels (lists) by default: PT_FLAG_NO_COOP / NO_LDS select the kernels that
    # walk the grid (LDS-resident and scalar-cache), FAST_RNG has its own instantiations of them
    for name, flags in (("grid, LDS", abi.PT_FLAG_NO_COOP), ("grid, scalar cache", abi.PT_FLAG_NO_LDS), ("default", 0),
                        ("grid, pixel-granular", abi.PT_FLAG_NO_COOP | abi.PT_FLAG_PIXEL_GRANULAR), ("stream", abi.PT_FLAG_FORCE_STREAM)):
        assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"sphere field seed {seed} {name}")
    F = abi.PT_FLAG_FAST_RNG
    assert_bit_identical(R.render_host(w, h, 70, ps, c, flags=F), orc.render(ps, c.c, w, h, 70, flags=F), f"sphere field seed {seed} fast mode")


def random_box_field(seed: int):
    """Runs of boxes for the exact slab culling (pt_device.hpp: box_run_culled), with what it has to get right varied at random:
    more than 16 boxes (chunks), thin slabs, boxes sharing faces / edges / corners (equal-t ties: the later one wins), exact
    duplicates, nested boxes (origins inside a box), stacks seen end-on (more than three candidates per ray: repeated passes),
    large and tiny coordinates, rect runs between box runs, mirrors and glass (rays that start on and inside boxes), and
    cameras inside, near and far."""
    rng = np.random.default_rng(seed)

    def color():
        return tuple(float(x) for x in rng.random(3))

    def material():
        k = rng.integers(0, 5)
        return (lambertian_material(color()) if k <= 1 else metal_material(color(), float(0.3 * rng.random())) if k == 2
                else dielectric_material(1.5, (1, 1, 1)) if k == 3 else lightsource_material(tuple(float(3 * x) for x in rng.random(3))))

    scale = float(10.0 ** rng.integers(-2, 4))          # scene size 0.01 .. 1000
    centre = np.array([(rng.random() - 0.5) * 2000 * scale * (seed % 4 == 1) for _ in range(3)])
    snap = scale / 8                                     # coordinates on a lattice: shared faces, edges and corners are exact
    hs = []

    def lattice_box():
        p0 = centre + np.round((rng.random(3) - 0.5) * scale / snap) * snap
        size = np.maximum(np.round(rng.random(3) * 0.35 * scale / snap), 1) * snap
        if rng.random() < 0.25:
            size[int(rng.integers(0, 3))] = snap * float(rng.choice([1e-3, 1 / 16]))   # a thin slab
        return box(tuple(p0), tuple(p0 + size), material())

    n = int(rng.integers(3, 40))
    for _ in range(n):
        hs.append(lattice_box())
        if rng.random() < 0.08:
            hs.append(hs[int(rng.integers(0, len(hs)))])                                  # exact duplicate
    if rng.random() < 0.5:                                                                # a stack seen end-on
        p0 = centre + np.array([0.3 * scale, 0, 0])
        for i in range(int(rng.integers(4, 9))):
            hs.append(box(tuple(p0 + [0, 0, i * snap]), tuple(p0 + [snap, snap, (i + 1) * snap]), material()))
    if rng.random() < 0.6:                                                                # a room around everything
        lo, hi = centre - 0.8 * scale, centre + 0.8 * scale
        hs.insert(0, box(tuple(lo - 0.01 * scale), (float(hi[0] + 0.01 * scale), float(lo[1]), float(hi[2] + 0.01 * scale)), material()))
        hs.append(box((float(lo[0] - 0.01 * scale), float(lo[1]), float(lo[2])), (float(lo[0]), float(hi[1]), float(hi[2])), material()))
    if rng.random() < 0.7:                                                                # rects between box runs
        k = int(rng.integers(1, len(hs)))
        a = centre - 0.5 * scale
        hs.insert(k, xy_rect(float(a[0]), float(a[0] + scale), float(a[1]), float(a[1] + scale), float(centre[2] + 0.5 * scale), material()))
        if rng.random() < 0.5:
            hs.insert(k, xz_rect(float(a[0]), float(a[0] + scale), float(a[2]), float(a[2] + scale), float(centre[1] - 0.5 * scale), material()))
    if rng.random() < 0.4:
        hs.append(sphere(tuple(centre), float(0.1 * scale), material()))
        hs.append(lattice_box())
        hs.append(lattice_box())
    dist = float(rng.choice([0.0, 0.2, 1.5, 30.0])) * scale
    frm = centre + np.array([dist * 0.6 + 0.01 * scale, dist * 0.45 + 0.013 * scale, dist * 0.65 + 0.017 * scale])
    cam = dict(look_from=tuple(float(x) for x in frm), look_at=tuple(float(x) for x in centre), vup=(0, 1, 0),
               vfov=float(70.0 if dist < scale else min(70.0, 2 * np.degrees(np.arctan(0.7 * scale / dist)) + 5.0)),
               aperture=0.0, focus_dist=float(max(dist, 0.1 * scale)), time0=0.0, time1=1.0)
    return pack(hs), cam


@pytest.mark.parametrize("seed", range(16))
def test_random_box_fields_through_the_slab_culling(orc, lib, seed):
    ps, cam = random_box_field(4000 + seed)
    w, h, spp = 40, 24, 8
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    # frames this small go to the cooperative kernels by default (straight-line box runs): NO_COOP / NO_LDS select the
    # kernels with the culled box runs (LDS-resident and scalar-cache)
    for name, flags in (("culled, LDS", abi.PT_FLAG_NO_COOP), ("culled, scalar cache", abi.PT_FLAG_NO_LDS), ("default", 0),
                        ("culled, pixel-granular", abi.PT_FLAG_NO_COOP | abi.PT_FLAG_PIXEL_GRANULAR), ("stream", abi.PT_FLAG_FORCE_STREAM),
                        ("plain division", abi.PT_FLAG_NO_FASTDIV | abi.PT_FLAG_NO_COOP)):
        assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"box field seed {seed} {name}")
    F = abi.PT_FLAG_FAST_RNG | abi.PT_FLAG_NO_COOP
    assert_bit_identical(R.render_host(w, h, 70, ps, c, flags=F), orc.render(ps, c.c, w, h, 70, flags=F), f"box field seed {seed} fast mode")


def random_triangle_field(seed: int, images: int = 0):
    """Long runs of Moller-Trumbore triangles for the exact triangle pool (csrc/pt_tripool.hpp; pt_device.hpp: tri_pool_scan),
    with what it has to get right varied at random: small random triangles, SLIVERS (edges nearly parallel: wide grazing
    bands, the always list), degenerate triangles (repeated vertices, collinear vertices), exact duplicates and coplanar
    re-orderings (equal-t ties: the later one wins), meshes with shared edges and vertices (rays through edges), triangles in
    the coordinate planes seen by cameras that look along those planes (every primary ray grazes them), huge and tiny
    scales, off-origin centres, a second triangle run behind another kind, and cameras inside, near, far and beyond the
    pool's rlimit (fallback to the full scan).  images = 1: an image-textured sphere beside the field (the pool kernels that carry
    the winner's u, v); images = 2: image-textured triangles as well (u, v of every accepted triangle)."""
    rng = np.random.default_rng(seed)
    atlas = TextureAtlas() if images else None
    img = image_texture.from_array(rng.integers(0, 256, (11, 19, 3), dtype=np.uint8), 1.5, atlas) if images else None

    def color():
        return tuple(float(x) for x in rng.random(3))

    def material():
        k = rng.integers(0, 6)
        return (lambertian_material(color()) if k <= 2 else metal_material(color(), float(0.3 * rng.random())) if k == 3
                else dielectric_material(1.5, (1, 1, 1)) if k == 4 else lightsource_material(tuple(float(3 * x) for x in rng.random(3))))

    scale = float(10.0 ** rng.integers(-2, 3))          # field size 0.01 .. 100
    centre = np.array([(rng.random() - 0.5) * 300 * scale * (seed % 4 == 1) for _ in range(3)])
    size = scale * float(rng.choice([0.03, 0.1, 0.3]))   # typical edge
    n = int(rng.integers(300, 2500))
    mats = [material() for _ in range(12)]
    if images >= 2:
        mats[0] = mats[5] = lambertian_material(img)
    hs = [sphere(tuple(centre + [0, -1000 * scale - 0.5 * scale, 0]), 1000 * scale, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    if images:
        hs.append(sphere(tuple(centre + [0.2 * scale, 0.1 * scale, -0.1 * scale]), 0.15 * scale, lambertian_material(img)))
    first = len(hs)

    def P():
        return centre + (rng.random(3) - 0.5) * scale

    def add(v0, v1, v2):
        hs.append(triangle(tuple(float(x) for x in v0), tuple(float(x) for x in v1), tuple(float(x) for x in v2), mats[int(rng.integers(0, len(mats)))]))

    while len(hs) - first < n:
        k = rng.integers(0, 12)
        v0 = P()
        if k <= 4:                                          # small random triangle
            add(v0, v0 + (rng.random(3) - 0.5) * size, v0 + (rng.random(3) - 0.5) * size)
        elif k == 5:                                        # sliver: second edge nearly parallel to the first
            e = (rng.random(3) - 0.5) * size
            add(v0, v0 + e, v0 + e * float(rng.random() * 1.5) + (rng.random(3) - 0.5) * size * float(10.0 ** rng.integers(-7, -2)))
        elif k == 6:                                        # degenerate: repeated or collinear vertices
            e = (rng.random(3) - 0.5) * size
            add(v0, v0 + e, v0 + e * 0.5) if rng.random() < 0.5 else add(v0, v0, v0 + e)
        elif k == 7 and len(hs) > first:                    # exact duplicate / same triangle with its vertices rotated or flipped
            t = hs[int(rng.integers(first, len(hs)))]
            if rng.random() < 0.5:
                hs.append(t)
            else:
                a, b, c_ = (np.array(t.v0), np.array(t.v1), np.array(t.v2)) if hasattr(t, "v0") else (v0, v0 + size, v0 - size)
                add(b, c_, a) if rng.random() < 0.5 else add(a, c_, b)
        elif k == 8:                                        # a strip of quads: shared edges and vertices, exactly
            u_, v_ = (rng.random(3) - 0.5) * size, (rng.random(3) - 0.5) * size
            for i in range(int(rng.integers(2, 7))):
                a = v0 + i * u_
                add(a, a + u_, a + v_)
                add(a + u_, a + u_ + v_, a + v_)
        elif k == 9:                                        # in a coordinate plane through the centre (lattice-snapped)
            ax = int(rng.integers(0, 3))
            a = v0.copy(); a[ax] = centre[ax]
            b = a + (rng.random(3) - 0.5) * size; b[ax] = centre[ax]
            c_ = a + (rng.random(3) - 0.5) * size; c_[ax] = centre[ax]
            add(a, b, c_)
        elif k == 10:                                       # a big one
            add(v0, v0 + (rng.random(3) - 0.5) * scale, v0 + (rng.random(3) - 0.5) * scale)
        else:                                               # tiny
            add(v0, v0 + (rng.random(3) - 0.5) * size * 1e-3, v0 + (rng.random(3) - 0.5) * size * 1e-3)
    if rng.random() < 0.6:                                  # interrupt the run; a second run (long or short) behind it
        p0 = centre + (rng.random(3) - 0.5) * scale * 0.5
        hs.append(box(tuple(p0), tuple(p0 + 0.1 * scale), material()))
        for _ in range(int(rng.choice([5, 400]))):
            v0 = P()
            add(v0, v0 + (rng.random(3) - 0.5) * size, v0 + (rng.random(3) - 0.5) * size)
    if rng.random() < 0.5:
        hs.append(xy_rect(float(centre[0] - scale), float(centre[0] + scale), float(centre[1] - scale), float(centre[1] + scale),
                          float(centre[2] - 0.7 * scale), lightsource_material((4, 4, 4))))
    mode = int(rng.integers(0, 5))
    dist = float([0.05, 0.6, 2.0, 40.0, 30000.0][mode]) * scale
    frm = centre + np.array([dist * 0.55, dist * 0.3, dist * 0.75])
    at = centre.copy()
    if seed % 3 == 0:                                       # look ALONG a coordinate plane through the centre
        frm = centre + np.array([dist + 0.3 * scale, 0.0, 0.0])
        at = centre + np.array([0.0, 0.0, 0.0])
    cam = dict(look_from=tuple(float(x) for x in frm), look_at=tuple(float(x) for x in at), vup=(0, 1, 0),
               vfov=float(70.0 if dist < scale else min(70.0, 2 * np.degrees(np.arctan(0.7 * scale / dist)) + 5.0)),
               aperture=0.0, focus_dist=float(max(dist, 0.1 * scale)), time0=0.0, time1=1.0)
    return (pack(hs, atlas) if images else pack(hs)), cam


@pytest.mark.parametrize("seed,images", [(8100, 1), (8101, 2), (8102, 2), (8104, 1), (8107, 2)])
def test_triangle_pool_with_image_textures(orc, lib, seed, images):
    """The triangle-pool kernels that carry texture coordinates: an image on a sphere beside the field (UV of the winner) and on
    triangles of the field itself (u, v of the accepted triangle: the pool's winner is known only when the scan is over)."""
    import ctypes as C
    ps, cam = random_triangle_field(seed, images)
    st = (C.c_int32 * 8)()
    w, h, spp = 40, 24, 8
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    with tri_pools():
        abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
        assert st[0] >= 300
        ds = R.DeviceScene(ps)
    for name, flags in (("pool", 0), ("full scan (stream)", abi.PT_FLAG_FORCE_STREAM)):
        assert_bit_identical(R.render_host(w, h, spp, ds, c, flags=flags), ref, f"textured triangle field seed {seed} images {images} {name}")
    F = abi.PT_FLAG_FAST_RNG
    assert_bit_identical(R.render_host(w, h, 70, ds, c, flags=F), orc.render(ps, c.c, w, h, 70, flags=F), f"textured triangle field seed {seed} fast mode")


@pytest.mark.parametrize("size", [(100, 37), (64, 64), (257, 9)])
def test_triangle_pool_scattered_pixels_cover_the_frame(orc, lib, size, monkeypatch):
    """The triangle-pool kernels deal consecutive queue positions to DIFFERENT tiles (pt_render.hip: lane_acquire, scatter_p): every
    pixel of a frame with padding tiles must still be rendered exactly once — whole frames and three shards, bit for bit, for
    single pixels (the default), runs of 4 and of 8 pixels, and with the scattering off."""
    ps, cam = random_triangle_field(8003)
    w, h = size
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, 3)
    for knob in (None, "2", "3", "off"):
        if knob == "off":
            monkeypatch.setenv("PT_NO_SCATTER", "1")
        elif knob is not None:
            monkeypatch.setenv("PT_SCATTER_LOG", knob)
        with tri_pools():
            ds = R.DeviceScene(ps)
        assert_bit_identical(R.render_host(w, h, 3, ds, c), ref, f"{w}x{h} scatter {knob}")
        for k in range(3):
            assert_bit_identical(R.render_host(w, h, 3, ds, c, shard_index=k, shard_count=3),
                                 orc.render(ps, c.c, w, h, 3, shard_index=k, shard_count=3), f"{w}x{h} scatter {knob}, shard {k}/3")


@pytest.mark.parametrize("seed", range(14))
def test_random_triangle_fields_through_the_triangle_pool(orc, lib, seed):
    ps, cam = random_triangle_field(8000 + seed)
    import ctypes as C
    st = (C.c_int32 * 8)()
    w, h, spp = 40, 24, 8
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    with tri_pools():
        abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
        assert st[0] >= 300, "the field's long run must get a pool"
        ds = R.DeviceScene(ps)
    for name, flags in (("pool", 0), ("pool, tile-granular", abi.PT_FLAG_TILE_GRANULAR), ("full scan (stream)", abi.PT_FLAG_FORCE_STREAM),
                        ("plain division: every ray irregular -> full scan in the pool kernel", abi.PT_FLAG_NO_FASTDIV)):
        assert_bit_identical(R.render_host(w, h, spp, ds, c, flags=flags), ref, f"triangle field seed {seed} {name}")
    F = abi.PT_FLAG_FAST_RNG
    assert_bit_identical(R.render_host(w, h, 70, ds, c, flags=F), orc.render(ps, c.c, w, h, 70, flags=F), f"triangle field seed {seed} fast mode")


@pytest.mark.parametrize("seed,tail", [(8000, "0"), (8001, "0.4"), (8003, "0"), (8004, "0.9"), (8006, "0.2"), (8009, "0"), (8012, "0.5")])
def test_binned_triangle_pool_renderer(orc, lib, seed, tail, monkeypatch):
    """PtTuning.tri_binned = 1 (csrc/pt_binned.hpp, opt-in): the frame in GENERATIONS — every live pixel one ray per generation, the rays' band
    stage brought together by direction bin (dense packets: 64 rays share a streamed list; sparse packets: the persistent kernel's own
    one-ray routine; rays outside the pool's domain: sliced full scans) — to the end of the frame (PT_BIN_TAIL=0) or with the tail handed to
    persistent waves at several points: the oracle's frame bit for bit, whole frames and shards, every ray irregular (PT_FLAG_NO_FASTDIV), and
    with packets forced dense (PT_BAND_DENSE_MIN=1) or sparse (=65)."""
    ps, cam = random_triangle_field(seed)
    w, h, spp = 40, 24, 8
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    monkeypatch.setenv("PT_BIN_TAIL", tail)
    ds = R.DeviceScene(ps, abi.tuning(tri_min_run=256, tri_binned=1))
    for dense_min in (None, "1", "65"):
        if dense_min is not None:
            monkeypatch.setenv("PT_BAND_DENSE_MIN", dense_min)
        assert_bit_identical(R.render_host(w, h, spp, ds, c), ref, f"binned renderer, field {seed}, tail {tail}, dense_min {dense_min}")
    monkeypatch.delenv("PT_BAND_DENSE_MIN")
    assert_bit_identical(R.render_host(w, h, spp, ds, c, flags=abi.PT_FLAG_NO_FASTDIV), ref, f"binned renderer, field {seed}: every ray outside the pool's domain")
    for k in range(3):
        assert_bit_identical(R.render_host(w, h, spp, ds, c, shard_index=k, shard_count=3),
                             orc.render(ps, c.c, w, h, spp, shard_index=k, shard_count=3), f"binned renderer, field {seed}, shard {k}/3")


@pytest.mark.parametrize("seed", [8000, 8002, 8005, 8011])
def test_camera_ray_candidate_cache_changes_no_bit(orc, lib, seed):
    """PtTuning.tri_cache (round 6): a lane keeps the grazing candidates of its pixel's camera rays — built once per pixel with the filters
    widened to the pixel's footprint — instead of enumerating the direction map for every sample.  With and without it, at frame sizes
    whose pixels are wide (every list overflows: no cache) and narrow (lists cached), many samples per pixel, whole frames and shards,
    and with a lens (the rays of a pixel no longer share their origin: no cache): the oracle's frame."""
    ps, cam = random_triangle_field(seed)
    orc.set_math(True)
    for (w, h, spp) in ((24, 16, 24), (320, 180, 3)):
        c = scenes.make_camera(cam, w, h)
        ref = orc.render(ps, c.c, w, h, spp)
        for cache in (0, -1):
            ds = R.DeviceScene(ps, abi.tuning(tri_min_run=256, tri_cache=cache))
            assert_bit_identical(R.render_host(w, h, spp, ds, c), ref, f"field {seed} {w}x{h}x{spp} tri_cache {cache}")
            if w == 24:
                assert_bit_identical(R.render_host(w, h, spp, ds, c, shard_index=1, shard_count=2),
                                     orc.render(ps, c.c, w, h, spp, shard_index=1, shard_count=2), f"field {seed} shard 1/2 tri_cache {cache}")
    lens = dict(cam, aperture=0.05)
    c = scenes.make_camera(lens, 64, 36)
    assert_bit_identical(R.render_host(64, 36, 6, R.DeviceScene(ps, abi.tuning(tri_min_run=256)), c), orc.render(ps, c.c, 64, 36, 6), f"field {seed} with a lens")


def test_triangle_pool_thresholds_and_small_runs(orc, lib, monkeypatch):
    """By default only runs of >= 4096 triangles get a pool (the fuzz fields do not: full scan); PT_TRICULL=1 lowers the threshold
    to 256; PT_NO_TRICULL overrides everything; PT_TRI_MIN=4 puts pools into the small mixed scenes (triangle runs of a handful,
    between spheres, boxes and media; Badouel-strategy scenes keep their kernels)."""
    import ctypes as C
    ps, cam = random_triangle_field(8003)
    st = (C.c_int32 * 8)()
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert list(st)[:6] == [0] * 6
    with tri_pools(PT_NO_TRICULL=1):
        abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
        assert list(st)[:6] == [0] * 6
    c = scenes.make_camera(cam, 40, 24)
    orc.set_math(True)
    assert_bit_identical(R.render_host(40, 24, 6, ps, c), orc.render(ps, c.c, 40, 24, 6), "default: no pool for a short run")
    for seed in (1001, 1003, 1004, 1006):
        ps, cam = random_scene(seed, allow_image_on_triangle=(seed % 2 == 0))
        c = scenes.make_camera(cam, 45, 27)
        ref = orc.render(ps, c.c, 45, 27, 12)
        with tri_pools(PT_TRI_MIN=4):
            ds = R.DeviceScene(ps)
        for flags in (0, abi.PT_FLAG_NO_LDS, abi.PT_FLAG_FORCE_STREAM):
            assert_bit_identical(R.render_host(45, 27, 12, ds, c, flags=flags), ref, f"PT_TRI_MIN=4 seed {seed} flags {flags}")


@pytest.mark.parametrize("knobs", [dict(tri_M=4.0), dict(tri_M=48.0), dict(tri_cell=0.6), dict(tri_cell=0.1), dict(tri_res=(32, 16, 16)),
                                   dict(tri_rho=(1.2, 2.5), tri_rho2=3.0),        # rays fall through all three classes ...
                                   dict(tri_rho=(-1.0, -1.0), tri_rho2=-1.0),     # ... or there is no map at all: every ray streams every band record
                                   dict(tri_budget_mb=1)])                          # maps over budget: built coarser or dropped
def test_triangle_pool_knobs_change_no_bit(orc, lib, knobs):
    """PtTuning's triangle-pool fields are performance-only (include/pt_render.h): the grid's slack 1 / M and cell size, the direction
    maps' resolutions, rho classes and memory budget — including the settings that leave rays without a map (they stream every band record)
    — give the image of the oracle's full scan, on a field with slivers, duplicates and degenerate triangles and from a camera inside it."""
    orc.set_math(True)
    for seed in (8001, 8004):
        ps, cam = random_triangle_field(seed)
        w, h, spp = 48, 27, 6
        c = scenes.make_camera(cam, w, h)
        ref = orc.render(ps, c.c, w, h, spp)
        t = abi.tuning(tri_min_run=256, **knobs)
        ds = R.DeviceScene(ps, t)
        assert_bit_identical(R.render_host(w, h, spp, ds, c), ref, f"triangle field {seed} with {knobs}")


def test_rays_that_graze_triangles(orc, lib):
    """Ray-level check aimed at the band: rays that lie ALMOST IN THE PLANE of a triangle of the 100 k-triangle mesh (tilted out of
    it by 1e-8 ... 1e-2) and pass near it — the rays for which the reference's binary32 test accepts triangles by rounding
    noise — from 0.01 to 12 units away, every one checked against the oracle's full scan (hit, t, scattered ray, RNG)."""
    ps, cam = scenes.build("triangles", n_triangles=100_000)
    hd = np.frombuffer(ps.hittables, dtype=scenes.hittable_dtype)
    f = hd["f"][1:-1].astype(np.float64)
    rng = np.random.default_rng(11)
    n = 24000
    recs = (abi.PtBounceIn * n)()
    for k in range(n):
        i = int(rng.integers(0, len(f)))
        v0, e1, e2 = f[i, 0:3], f[i, 3:6] - f[i, 0:3], f[i, 6:9] - f[i, 0:3]
        nn = np.cross(e1, e2)
        nn /= max(np.linalg.norm(nn), 1e-30)
        t1 = e1 / max(np.linalg.norm(e1), 1e-30)
        t2 = np.cross(nn, t1)
        ang = rng.uniform(0, 2 * np.pi)
        d = np.cos(ang) * t1 + np.sin(ang) * t2 + nn * (10.0 ** rng.uniform(-8, -2)) * rng.choice([-1.0, 1.0])
        d *= rng.uniform(0.3, 2.0)
        target = v0 + rng.uniform(-0.2, 1.2) * e1 + rng.uniform(-0.2, 1.2) * e2 + rng.normal(size=3) * 10.0 ** rng.uniform(-7, -3)
        o = target - d / np.linalg.norm(d) * rng.uniform(0.01, 12.0)
        recs[k].origin[:] = [float(x) for x in o]; recs[k].dir[:] = [float(x) for x in d]; recs[k].time = 0.5
        recs[k].rng_state = int(rng.integers(1, 2 ** 32)); recs[k].attenuation[:] = [1.0, 1.0, 1.0]
    with tri_pools():
        ds = R.DeviceScene(ps)
    out = (abi.PtBounceOut * n)()
    abi.check(lib.pt_debug_bounce(ds.handle, recs, out, n), "pt_debug_bounce")
    orc.set_math(True)
    ref = orc.bounce(ps, recs)
    bad = [k for k in range(n) if (out[k].status, out[k].hittable, np.float32(out[k].t).tobytes(), out[k].rng_state)
           != (ref[k].status, ref[k].hittable, np.float32(ref[k].t).tobytes(), ref[k].rng_state)]
    assert not bad, f"{len(bad)} of {n} grazing rays differ; first: ray {bad[0]} device {out[bad[0]].hittable} t {out[bad[0]].t!r} oracle {ref[bad[0]].hittable} t {ref[bad[0]].t!r}"
    hits = sum(1 for k in range(n) if ref[k].status != abi.PT_BOUNCE_MISS)
    assert hits > n // 2


@pytest.mark.parametrize("kind,seed", [("box", 4001), ("box", 4004), ("box", 4005), ("box", 4012), ("box", 4015), ("sphere", 3001),
                                       ("sphere", 3007), ("random", 1003), ("random", 1004), ("triangle", 8001), ("triangle", 8002), ("triangle", 8006)])
def test_path_rays_through_fuzz_scenes(orc, lib, kind, seed):
    """Paths followed with the oracle, the device checked on every ray of every generation (tests/path_rays.py): 15 000 camera
    rays and everything they scatter into — rays that start ON faces, in glass, next to shared faces and duplicates.  (The
    slab pool's key accounting once lost a fourth candidate after a dropped key: three framebuffer fuzz suites did not see it,
    this did within 100 000 rays.)"""
    from path_rays import follow_paths
    ps, cam = {"box": random_box_field, "sphere": random_sphere_field, "random": lambda s: random_scene(s, False),
               "triangle": random_triangle_field}[kind](seed)
    c = scenes.make_camera(cam, 40, 24)
    with forced_pools() if kind == "random" else tri_pools() if kind == "triangle" else contextlib.nullcontext():
        checked, bad = follow_paths(lib, orc, ps, c.c, 40, 24, 15000, 12, seed)
    assert checked >= 15000 and not bad, f"{len(bad)} of {checked} rays differ: " + " | ".join(bad[:3])


@pytest.mark.parametrize("kind,seed", [("sphere", 3001), ("sphere", 3004), ("sphere", 3007), ("sphere", 3010), ("smoke", 0)])
def test_path_rays_through_the_queued_sphere_grid_walk(orc, lib, kind, seed, monkeypatch):
    """The same ray-level check with PtTuning.grid_walk = 2 (PT_GRID_WALK): the probe kernel then walks sphere grids through the LDS pair
    queue (64 (ray, sphere) pairs per batch, one 64-bit LDS minimum per hit carrying the tie rule) like the render kernels the launcher
    picks for divergent or chain-bound launches — on sphere fields (duplicates, moving spheres, spheres that touch) and on the
    496-hittable scene itself."""
    from path_rays import follow_paths
    monkeypatch.setenv("PT_GRID_WALK", "2")
    if kind == "smoke":
        ps, cam = scenes.build("smoke", textures="procedural")
    else:
        ps, cam = random_sphere_field(seed)
    c = scenes.make_camera(cam, 40, 24)
    checked, bad = follow_paths(lib, orc, ps, c.c, 40, 24, 15000, 12, seed + 5)
    assert checked >= 15000 and not bad, f"{len(bad)} of {checked} rays differ: " + " | ".join(bad[:3])


@pytest.mark.parametrize("name,n,gens", [("cornell", 40000, 16), ("smoke", 20000, 10), ("triangles", 6000, 6)])
def test_path_rays_on_the_baseline_scenes(orc, lib, name, n, gens):
    """The same ray-level check on the scenes BASELINE.json names: the Cornell-style scene (slab pool: every ray of every
    generation starts on a face of one of its boxes) and the 496-hittable scene (sphere grid, image textures, media)."""
    from path_rays import follow_paths
    ps, cam = scenes.build(name, **({"n_triangles": 100_000} if name == "triangles" else {}))
    w, h = (192, 108) if name == "cornell" else (200, 112)
    c = scenes.make_camera(cam, w, h)
    with tri_pools() if name == "triangles" else contextlib.nullcontext():  # the 100 k-triangle mesh through its triangle pool
        checked, bad = follow_paths(lib, orc, ps, c.c, w, h, n, gens, 7)
    assert checked >= 1.5 * n and not bad, f"{len(bad)} of {checked} rays differ: " + " | ".join(bad[:3])


@pytest.mark.gpu
def test_scheduling_paths_change_no_frame():
    """Round 5's scheduling — the cost probe whose samples are kept, the dilated cost map, the heaviest-first order, the issue priorities of the
    headline family — decides only WHEN a pixel is rendered.  Random (scene, frame size, samples, shard): the default path's framebuffer must
    equal, bit for bit, the one rendered with none of it (PT_FLAG_NO_LPT; PtTuning.chain_priority = probe_resume = -1).  The short form of
    tools/soak_scheduling.py (profiles/r05_soak_scheduling.log: 300 frames); the parity tests compare the default path with the oracle."""
    import torch
    import scenes_small as S
    rng = np.random.default_rng(17)
    pool = {"cornell": scenes.build("cornell"), "smoke": scenes.build("smoke"), "field": S.sphere_field_scene(), "mixed": S.mixed_scene()}
    plain = abi.tuning(chain_priority=-1, probe_resume=-1)
    made = {}
    for case in range(28):
        name = list(pool)[case % 4]
        ps, cam_args = pool[name]
        W, H = int(rng.integers(64, 800)), int(rng.integers(64, 500))
        spp = int(rng.choice([16, 17, 32, 48, 96]))
        n = int(rng.choice([1, 1, 2, 3]))
        idx = int(rng.integers(0, n))
        cam = scenes.make_camera(cam_args, W, H)
        if name not in made:
            made[name] = (R.DeviceScene(ps), R.DeviceScene(ps, tuning=plain))
        a = R.render(W, H, spp, made[name][0], cam, shard_index=idx, shard_count=n)
        b = R.render(W, H, spp, made[name][1], cam, shard_index=idx, shard_count=n, flags=abi.PT_FLAG_NO_LPT)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (name, W, H, spp, idx, n)
