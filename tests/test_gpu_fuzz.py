"""Randomised scenes (every hittable kind, material and texture, random order so that kinds interleave in short runs,
media anywhere in the list, coincident surfaces) rendered by every kernel flavour and compared with the oracle
bit for bit."""
import numpy as np
import pytest

from conftest import assert_bit_identical
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
from path_tracer_amd.scene import (TextureAtlas, box, checker_texture, constant_medium, dielectric_material,
                                   image_texture, lambertian_material, lightsource_material, metal_material, pack,
                                   sphere, triangle, xy_rect, xz_rect, yz_rect)

pytestmark = pytest.mark.gpu


def random_scene(seed: int, allow_image_on_triangle: bool):
    rng = np.random.default_rng(seed)
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
            hs.append(triangle(tuple(v0), tuple(v0 + (rng.random(3) - 0.5)), tuple(v0 + (rng.random(3) - 0.5)), material(True)))
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
