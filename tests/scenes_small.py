"""Small scenes shared by the CPU and GPU tests: every hittable kind, every material, every texture,
the edge cases the reference's semantics make interesting (ties, nested media, moving spheres, stale UV)."""
import numpy as np

from path_tracer_amd import scenes
from path_tracer_amd.scene import (TextureAtlas, box, checker_texture, constant_medium, dielectric_material,
                                   image_texture, lambertian_material, lightsource_material, metal_material, pack,
                                   sphere, triangle, xy_rect, xz_rect, yz_rect)


def _atlas_image(atlas, w=37, h=23, freq=1.0):
    y, x = np.mgrid[0:h, 0:w]
    rgb = np.stack([(x * 7 + y * 3) % 256, (x * 5 + 11 * y) % 256, (x * y) % 256], axis=-1).astype(np.uint8)
    return image_texture.from_array(rgb, freq, atlas)


def mixed_scene():
    """All seven hittable kinds, five materials, three textures, both medium boundaries, in an order that
    interleaves kinds (many short runs)."""
    atlas = TextureAtlas()
    img = _atlas_image(atlas)
    img5 = _atlas_image(atlas, 16, 9, 5.0)
    hs = [
        sphere((0, -100.5, -1), 100, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))),
        sphere((0, 0, -1), 0.5, lambertian_material((0.7, 0.3, 0.3))),
        xy_rect(-2, -1, -0.5, 1, -1.5, lambertian_material(img)),
        sphere((1, 0, -1), 0.5, metal_material((0.8, 0.6, 0.2), 0.3)),
        sphere((-1, 0, -1), 0.5, dielectric_material(1.5, (1, 1, 1))),
        sphere((-1, 0, -1), -0.45, dielectric_material(1.5, (1.0, 0.9, 0.9))),  # negative radius: hollow glass
        triangle((-0.5, 0.6, -1.2), (0.5, 0.6, -1.2), (0, 1.3, -0.9), lambertian_material(img5)),  # stale-UV path
        box((1.2, -0.5, -2.5), (1.8, 0.7, -1.9), metal_material((0.7, 0.6, 0.5), 0.0)),
        sphere((0.3, 0.1, -0.2), (0.3, 0.3, -0.2), 0.0, 1.0, 0.12, lambertian_material(img)),  # moving + image
        constant_medium(sphere((0.8, 0.9, -1.2), 0.4, lambertian_material((1, 1, 1))), 3.0, (0.9, 0.9, 1.0)),
        xz_rect(-1, 1, -2, 0, 2.5, lightsource_material((4, 4, 4))),
        yz_rect(-0.5, 1.5, -2.5, -0.5, -2.2, lambertian_material((0.2, 0.8, 0.2))),
        constant_medium(box((-1.9, -0.5, -0.9), (-1.3, 0.2, -0.3), lambertian_material((1, 1, 1))), 5.0,
                        checker_texture((0.1, 0.1, 0.1), (0.9, 0.2, 0.2))),
        sphere((0, 0.2, 0.6), 0.15, lightsource_material(img5)),
    ]
    cam = dict(look_from=(0.3, 0.6, 2.5), look_at=(0, 0.2, -1), vup=(0, 1, 0), vfov=50.0, aperture=0.1,
               focus_dist=3.4, time0=0.0, time1=1.0)
    return pack(hs, atlas), cam


def spheres_scene():
    """No image texture, no medium: spheres (static + moving), all scattering materials, checker."""
    hs = [sphere((0, -1000, 0), 1000, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    rng = scenes.HostRNG(4242)
    for a in range(-3, 3):
        for b in range(-3, 3):
            c = (a + 0.9 * float(rng.float_t()), 0.2, b + 0.9 * float(rng.float_t()))
            m = float(rng.float_t())
            if m < 0.3:
                hs.append(sphere(c, 0.2, lambertian_material(tuple(rng.vec_t()))))
            elif m < 0.6:
                c2 = (c[0], c[1] + 0.3 * float(rng.float_t()), c[2])
                hs.append(sphere(c, c2, 0.0, 1.0, 0.2, lambertian_material(tuple(rng.vec_t()))))
            elif m < 0.8:
                hs.append(sphere(c, 0.2, metal_material(tuple(rng.vec_t(0.5, 1)), float(rng.float_t(0, 0.5)))))
            else:
                hs.append(sphere(c, 0.2, dielectric_material(1.5, (1, 1, 1))))
    hs.append(sphere((0, 1, 0), 1.0, dielectric_material(1.5, (1.0, 0.5, 0.5))))
    hs.append(sphere((4, 1, 0), 0.2, lightsource_material((10, 0, 10))))
    cam = dict(look_from=(6, 2, 3), look_at=(0, 0.5, 0), vup=(0, 1, 0), vfov=40.0, aperture=0.05, focus_dist=7.0,
               time0=0.0, time1=1.0)
    return pack(hs), cam


def triangles_scene(n=300):
    return scenes.triangle_mesh_scene(n, seed=7, n_colors=8)


def ties_scene():
    """Coplanar / touching surfaces: equal-t ties resolve by list position (rects accept t == max,
    spheres need t < max: rectangle.hpp:36 vs sphere.hpp:77)."""
    white = lambertian_material((0.8, 0.8, 0.8))
    red = lambertian_material((0.9, 0.1, 0.1))
    blue = lambertian_material((0.1, 0.1, 0.9))
    hs = [
        xy_rect(-1, 1, -1, 1, -2, red),
        xy_rect(-0.5, 1.5, -0.5, 1.5, -2, blue),          # same plane, later in the list: wins the overlap
        box((-2, -1.5, -3), (2, -1, -1), white),
        box((-2, -1.5, -3), (0, -1, -1), red),            # shares faces with the previous box
        sphere((0, 0, -2), 0.5, white),
        sphere((0, 0, -2), 0.5, blue),                    # identical sphere later: loses (strict <)
        triangle((-1.5, 1, -2), (-0.5, 1, -2), (-1, 1.8, -2), red),
        triangle((-1.5, 1, -2), (-0.5, 1, -2), (-1, 1.8, -2), blue),
    ]
    cam = dict(look_from=(0, 0, 1), look_at=(0, 0, -2), vup=(0, 1, 0), vfov=70.0, aperture=0.0, focus_dist=3.0,
               time0=0.0, time1=0.0)
    return pack(hs), cam


def sphere_ties_scene():
    """Equal-t ties between spheres that the device scans out of list order (static spheres of a run before its moving
    ones): duplicates of one sphere as static / "moving" with center1 == center0 (same t at every ray time) in both list
    orders, next to ordinary static and moving neighbours.  The reference keeps the FIRST in list order (sphere.hpp:77)."""
    red, green, blue, white = (lambertian_material(c) for c in ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.8, 0.8, 0.8)))
    light = lightsource_material((3, 3, 3))
    hs = [
        sphere((0, -100.5, -2), 100, white),
        sphere((-1.2, 0, -2), (-1.2, 0, -2), 0.0, 1.0, 0.5, red),   # "moving", not displaced: first in list -> wins
        sphere((-1.2, 0, -2), 0.5, green),                           # static duplicate, scanned FIRST on the device
        sphere((0, 0, -2), 0.5, blue),                               # static first in list -> wins
        sphere((0, 0, -2), (0, 0, -2), 0.0, 1.0, 0.5, red),          # "moving" duplicate
        sphere((0, 0, -2), 0.5, green),                              # second static duplicate
        sphere((1.2, 0, -2), (1.2, 0.3, -2), 0.0, 1.0, 0.5, green),  # really moving
        sphere((1.2, 0, -2), (1.2, 0.3, -2), 0.0, 1.0, 0.5, light),  # its duplicate: loses
        sphere((0.3, 0.9, -1.6), 0.25, metal_material((0.8, 0.8, 0.8), 0.1)),
        sphere((-0.5, 0.8, -1.5), (-0.5, 1.0, -1.5), 0.0, 1.0, 0.2, dielectric_material(1.5, (1, 1, 1))),
    ]
    cam = dict(look_from=(0, 0.4, 1.5), look_at=(0, 0.1, -2), vup=(0, 1, 0), vfov=60.0, aperture=0.0, focus_dist=3.5,
               time0=0.0, time1=1.0)
    return pack(hs), cam


def absorbed_ties_scene():
    """Equal-t ties across ABSORBED sphere runs (round 6, pt_flatten.hpp): run 0 (four spheres: lists) also tests the static spheres of
    the two later sphere runs that only a rect, three boxes (a slab pool) and a triangle separate it from, i.e. EARLIER than the reference's
    list order does.  Surfaces are placed so that axis-parallel rays tie exactly (a sphere's pole in a rect's / a box face's / a triangle's
    plane: t = 2.5 from z = 1) — with a sphere of run 0 (it loses to the later accept-equal kinds), with a sphere of a LATER run only (the
    rect / box / triangle come first in the list and keep the hit: the device, which tested the sphere first, must let them replace it) —
    and spheres duplicated across the runs (the first in list order wins)."""
    red, green, blue, white = (lambertian_material(c) for c in ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.8, 0.8, 0.8)))
    light = lightsource_material((3, 3, 3))
    hs = [
        sphere((0, -100.5, -2), 100, white),
        sphere((0, 0, -2), 0.5, blue),                                     # A: its front pole (0, 0, -1.5) lies in the rect's plane
        sphere((0, 1.1, -2), 0.3, red),
        sphere((-2.4, 0, -2), 0.5, green),                                 # F: duplicated in the last run
        xy_rect(-0.3, 0.3, -0.3, 0.3, -1.5, white),                        # ties with A (earlier: the rect wins) and with A' (later: the rect wins)
        box((0.7, -0.5, -2.5), (1.7, 0.5, -1.5), red),                     # its front face ties with C' (a later run's sphere)
        box((0.9, 0.5, -2.3), (1.5, 0.9, -1.7), green),
        box((-3.4, -0.5, -2.5), (-2.9, 0.5, -1.5), blue),
        sphere((0, 0, -2), 0.5, green),                                    # A': duplicate of A, later: loses to A (and to the rect)
        sphere((1.2, 0, -2), 0.5, light),                                  # C': front pole (1.2, 0, -1.5) in the box's front face; the box is earlier: wins
        triangle((-1.6, -0.4, -1.5), (-0.8, -0.4, -1.5), (-1.2, 0.5, -1.5), white),  # ties with B' below (later): the triangle wins
        sphere((-1.2, 0, -2), 0.5, red),                                   # B': front pole (-1.2, 0, -1.5) in the triangle's plane
        sphere((-2.4, 0, -2), 0.5, blue),                                  # F': duplicate of F (run 0): F wins
        sphere((0.3, 0.9, -1.6), 0.2, metal_material((0.8, 0.8, 0.8), 0.1)),
    ]
    cam = dict(look_from=(0, 0.3, 1.5), look_at=(0, 0.1, -2), vup=(0, 1, 0), vfov=75.0, aperture=0.0, focus_dist=3.5,
               time0=0.0, time1=0.0)
    return pack(hs), cam


def badouel_scene():
    """Triangles with the Badouel intersection strategy (triangle.hpp:14-56, `_triangle<badouel_ray_triangle_intersec>`) next to
    Moller-Trumbore ones and other kinds: separate runs in list order, a shared edge, a coplanar duplicate pair (the later
    one wins the tie: both strategies accept t == max), a sliver, one behind a glass sphere."""
    red, green, blue, white = (lambertian_material(c) for c in ((0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.8, 0.8, 0.8)))
    B = "badouel"
    hs = [
        sphere((0, -100.5, -2), 100, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))),
        triangle((-1.6, -0.4, -2.0), (-0.4, -0.4, -2.0), (-1.0, 0.8, -2.2), red, B),
        triangle((-0.4, -0.4, -2.0), (0.8, -0.4, -2.0), (-1.0, 0.8, -2.2), green, B),       # shares an edge with the first
        triangle((0.2, 0.0, -1.6), (1.2, 0.0, -1.6), (0.7, 0.9, -1.6), blue),                 # Moller-Trumbore
        triangle((0.2, 0.0, -1.6), (1.2, 0.0, -1.6), (0.7, 0.9, -1.6), red, B),               # the same triangle, Badouel, later in the list
        triangle((-0.2, 0.9, -1.8), (0.2, 0.9, -1.8), (0.0, 0.9001, -1.2), metal_material((0.8, 0.8, 0.8), 0.1), B),  # sliver
        sphere((-0.9, 0.1, -1.2), 0.35, dielectric_material(1.5, (1, 1, 1))),
        triangle((-2.0, 1.2, -3.0), (2.0, 1.2, -3.0), (0.0, 2.4, -2.0), lightsource_material((3, 3, 2.5)), B),
        box((1.3, -0.5, -2.4), (1.8, 0.2, -1.9), white),
    ]
    cam = dict(look_from=(0.1, 0.5, 1.2), look_at=(0, 0.2, -2), vup=(0, 1, 0), vfov=65.0, aperture=0.02, focus_dist=3.2,
               time0=0.0, time1=1.0)
    return pack(hs), cam


def sphere_field_scene(n=260, seed=31):
    """Enough small spheres for the culling grid of pt_flatten.hpp (>= 48): a slab of static and moving ones (some touching,
    some duplicated for equal-t ties, some overlapping cell borders), big spheres that stay outside the grid (ground, a glass
    ball INSIDE the field, a mirror ball), a hollow (negative-radius) small sphere, and other kinds after the run."""
    rng = scenes.HostRNG(seed)
    hs = [sphere((0, -500, 0), 500, lambertian_material(checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))))]
    for i in range(n):
        c = (-6 + 12 * float(rng.float_t()), 0.15 + 0.5 * float(rng.float_t()), -6 + 12 * float(rng.float_t()))
        r = 0.12 + 0.1 * float(rng.float_t())
        k = float(rng.float_t())
        mat = (lambertian_material(tuple(rng.vec_t())) if k < 0.5 else metal_material(tuple(rng.vec_t(0.5, 1)), 0.3 * float(rng.float_t()))
               if k < 0.8 else dielectric_material(1.5, (1, 1, 1)))
        if i % 3 == 1:
            hs.append(sphere(c, (c[0] + 0.2 * float(rng.float_t()), c[1] + 0.4 * float(rng.float_t()), c[2]), 0.0, 1.0, r, mat))
        else:
            hs.append(sphere(c, r, mat))
        if i % 41 == 7:
            hs.append(sphere(c, r, lightsource_material((2, 2, 2))))  # exact duplicate later in the list: loses every tie
    hs.append(sphere((1.0, 0.6, 0.5), -0.18, dielectric_material(1.5, (1, 1, 1))))   # hollow small sphere
    hs.append(sphere((0.0, 1.0, 0.0), 1.0, dielectric_material(1.5, (1.0, 0.8, 0.8))))  # big, in the middle of the field
    hs.append(sphere((-3.0, 1.2, 2.0), 1.2, metal_material((0.8, 0.8, 0.9), 0.0)))
    hs.append(triangle((2, 0, 2), (3, 0, 2), (2.5, 1.2, 2.2), lambertian_material((0.9, 0.2, 0.2))))
    hs.append(sphere((4, 2.5, -1), 0.3, lightsource_material((8, 8, 6))))
    hs.append(constant_medium(sphere((-2, 0.8, -2), 0.9, lambertian_material((1, 1, 1))), 1.5, (0.9, 0.9, 1.0)))
    cam = dict(look_from=(9, 3.5, 7), look_at=(0, 0.3, 0), vup=(0, 1, 0), vfov=40.0, aperture=0.05, focus_dist=11.0,
               time0=0.0, time1=1.0)
    return pack(hs), cam


def empty_scene():
    cam = dict(look_from=(0, 0, 1), look_at=(0, 0, -1), vup=(0, 1, 0), vfov=60.0, aperture=0.0, focus_dist=1.0,
               time0=0.0, time1=0.0)
    return pack([]), cam


def cornell_scene():
    return scenes.build("cornell")


ALL = {"cornell": cornell_scene, "mixed": mixed_scene, "spheres": spheres_scene, "triangles": triangles_scene,
       "ties": ties_scene, "sphere_ties": sphere_ties_scene, "absorbed_ties": absorbed_ties_scene, "badouel": badouel_scene, "sphere_field": sphere_field_scene, "empty": empty_scene}
