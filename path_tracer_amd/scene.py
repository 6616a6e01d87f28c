"""Host-side mirror of the reference's scene-description types (Python flavour).

Same names, argument order and meaning as the reference constructors so scene code reads
like the reference's `main.cpp` (all citations into /root/reference/include):

    sphere(cen, r, mat) / sphere(cen0, cen1, t0, t1, r, mat)      sphere.hpp:30,40
    xy_rect(x0,x1,y0,y1,k,mat)  xz_rect  yz_rect                   rectangle.hpp:21,59,97
    triangle(v0,v1,v2,mat)                                          triangle.hpp:107
    box(p0,p1,mat)                                                  box.hpp:15
    constant_medium(boundary, density, color|texture)               constant_medium.hpp:18,23
    lambertian_material(color|texture)  metal_material(color,fuzz)  dielectric_material(ri,albedo)
    lightsource_material(color|texture) isotropic_material(color|texture)   material.hpp:11-131
    solid_texture(color)  checker_texture(odd, even)  image_texture.image_texture_factory(path, freq)
                                                                    texture.hpp:18-152
    camera(look_from, look_at, vup, vfov, aspect, aperture, focus_dist, t0=0, t1=0)   camera.hpp:67-69

`pack(hittables)` turns a list of those values (the reference's std::vector<hittable_t>) into the
C-ABI tables of include/pt_render.h.  List order is preserved: it is traversal order.
All scalars are rounded to binary32 on construction, as the reference's `real_t` fields are.
"""
from __future__ import annotations

import ctypes as C
import sys
from dataclasses import dataclass, field
from typing import Iterable, Sequence

import numpy as np

from . import abi

f32 = np.float32


def _c3(c) -> tuple:
    a = np.asarray(c, dtype=np.float32).reshape(3)
    return (float(a[0]), float(a[1]), float(a[2]))


# ---- textures (texture.hpp) ---------------------------------------------------------------

@dataclass(frozen=True)
class solid_texture:
    color: tuple

    def __init__(self, *c):
        object.__setattr__(self, "color", _c3(c[0] if len(c) == 1 else c))


@dataclass(frozen=True)
class checker_texture:
    """First argument is the colour used where sin*sin*sin < 0 (`odd`, texture.hpp:38-40)."""
    odd: tuple
    even: tuple

    def __init__(self, odd, even):
        object.__setattr__(self, "odd", odd.color if isinstance(odd, solid_texture) else _c3(odd))
        object.__setattr__(self, "even", even.color if isinstance(even, solid_texture) else _c3(even))


class TextureAtlas:
    """The serialized RGB8 store behind image_texture (texture.hpp:71,113-114,157): starts with the
    {0,0,1} fallback texel; every image is appended rows-top-down and addressed by its texel offset."""

    def __init__(self):
        self.data = bytearray([0, 0, 1])

    def append(self, rgb: np.ndarray) -> int:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("image must be [height][width][3] uint8")
        offset = len(self.data) // 3
        self.data += rgb.tobytes()
        return offset

    def bytes(self) -> bytes:
        return bytes(self.data)


default_atlas = TextureAtlas()


@dataclass(frozen=True)
class image_texture:
    width: int
    height: int
    offset: int
    cyclic_frequency: float
    atlas: TextureAtlas = field(compare=False, hash=False, repr=False, default=None)

    @staticmethod
    def from_array(rgb: np.ndarray, cyclic_frequency: float = 1.0, atlas: TextureAtlas | None = None) -> "image_texture":
        atlas = atlas or default_atlas
        off = atlas.append(rgb)
        return image_texture(int(rgb.shape[1]), int(rgb.shape[0]), off, float(f32(cyclic_frequency)), atlas)

    @staticmethod
    def image_texture_factory(file_name: str, cyclic_frequency: float = 1.0,
                              atlas: TextureAtlas | None = None) -> "image_texture":
        """texture.hpp:97-117.  A load failure prints to stderr and yields the 1x1 texture at offset 0."""
        atlas = atlas or default_atlas
        try:
            from PIL import Image  # decoder stands in for stb_image (absent)
            with Image.open(file_name) as im:
                rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        except Exception as e:  # noqa: BLE001 - mirror the reference: report and fall back
            print(f"ERROR: Could not load texture image file '{file_name}'.\n{e}", file=sys.stderr)
            return image_texture(1, 1, 0, float(f32(cyclic_frequency)), atlas)
        return image_texture.from_array(rgb, cyclic_frequency, atlas)


def _as_texture(a):
    if isinstance(a, (solid_texture, checker_texture, image_texture)):
        return a
    return solid_texture(a)


# ---- materials (material.hpp) ------------------------------------------------------------------

@dataclass(frozen=True)
class lambertian_material:
    albedo: object

    def __init__(self, a):
        object.__setattr__(self, "albedo", _as_texture(a))


@dataclass(frozen=True)
class metal_material:
    albedo: tuple
    fuzz: float

    def __init__(self, a, f):
        object.__setattr__(self, "albedo", _c3(a))
        object.__setattr__(self, "fuzz", float(min(max(f32(f), f32(0.0)), f32(1.0))))  # std::clamp, material.hpp:37


@dataclass(frozen=True)
class dielectric_material:
    ref_idx: float
    albedo: tuple

    def __init__(self, ri, albedo):
        object.__setattr__(self, "ref_idx", float(f32(ri)))
        object.__setattr__(self, "albedo", _c3(albedo))


@dataclass(frozen=True)
class lightsource_material:
    emit: object

    def __init__(self, a):
        object.__setattr__(self, "emit", _as_texture(a))


@dataclass(frozen=True)
class isotropic_material:
    albedo: object

    def __init__(self, a):
        object.__setattr__(self, "albedo", _as_texture(a))


# ---- hittables ------------------------------------------------------------------------------------

@dataclass(frozen=True)
class sphere:
    center0: tuple
    center1: tuple
    radius: float
    time0: float
    time1: float
    material_type: object

    def __init__(self, *args):
        if len(args) == 3:  # sphere(cen, r, mat)  sphere.hpp:30-36
            cen, r, mat = args
            c0 = c1 = _c3(cen)
            t0 = t1 = 0.0
        elif len(args) == 6:  # sphere(cen0, cen1, t0, t1, r, mat)  sphere.hpp:40-47
            cen0, cen1, t0, t1, r, mat = args
            c0, c1 = _c3(cen0), _c3(cen1)
        else:
            raise TypeError("sphere(cen, r, mat) or sphere(cen0, cen1, time0, time1, r, mat)")
        for k, v in (("center0", c0), ("center1", c1), ("radius", float(f32(r))), ("time0", float(f32(t0))),
                     ("time1", float(f32(t1))), ("material_type", mat)):
            object.__setattr__(self, k, v)


@dataclass(frozen=True)
class _rect:
    a0: float
    a1: float
    b0: float
    b1: float
    k: float
    material_type: object

    def __init__(self, a0, a1, b0, b1, k, mat):
        for name, v in zip(("a0", "a1", "b0", "b1", "k"), (a0, a1, b0, b1, k)):
            object.__setattr__(self, name, float(f32(v)))
        object.__setattr__(self, "material_type", mat)


class xy_rect(_rect):
    """xy_rect(x0, x1, y0, y1, k, mat) rectangle.hpp:21"""


class xz_rect(_rect):
    """xz_rect(x0, x1, z0, z1, k, mat) rectangle.hpp:59 — top-level use is an extension (render.hpp:22-23)."""


class yz_rect(_rect):
    """yz_rect(y0, y1, z0, z1, k, mat) rectangle.hpp:97 — top-level use is an extension."""


@dataclass(frozen=True)
class triangle:
    """triangle.hpp:102-122.  `strategy` mirrors the class template's argument: "moller_trumbore" (the reference's
    `triangle` alias, what main.cpp builds) or "badouel" (`_triangle<badouel_ray_triangle_intersec>`, triangle.hpp:14-56)."""
    v0: tuple
    v1: tuple
    v2: tuple
    material_type: object
    strategy: str = "moller_trumbore"

    def __init__(self, v0, v1, v2, mat, strategy: str = "moller_trumbore"):
        if strategy not in ("moller_trumbore", "badouel"):
            raise ValueError("triangle strategy must be 'moller_trumbore' or 'badouel'")
        object.__setattr__(self, "v0", _c3(v0))
        object.__setattr__(self, "v1", _c3(v1))
        object.__setattr__(self, "v2", _c3(v2))
        object.__setattr__(self, "material_type", mat)
        object.__setattr__(self, "strategy", strategy)


@dataclass(frozen=True)
class box:
    box_min: tuple
    box_max: tuple
    material_type: object

    def __init__(self, p0, p1, mat):
        object.__setattr__(self, "box_min", _c3(p0))
        object.__setattr__(self, "box_max", _c3(p1))
        object.__setattr__(self, "material_type", mat)


@dataclass(frozen=True)
class constant_medium:
    boundary: object
    neg_inv_density: float
    phase_function: isotropic_material

    def __init__(self, b, d, a):
        if not isinstance(b, (sphere, box)):
            raise TypeError("constant_medium boundary must be a sphere or a box (constant_medium.hpp:10)")
        object.__setattr__(self, "boundary", b)
        object.__setattr__(self, "neg_inv_density", float(f32(-1.0) / f32(d)))  # constant_medium.hpp:20
        object.__setattr__(self, "phase_function", isotropic_material(a))


# ---- camera (camera.hpp:67-87) -----------------------------------------------------------------------

class camera:
    def __init__(self, look_from, look_at, vup, degree_vfov, aspect_ratio, aperture, focus_dist, time0=0.0, time1=0.0):
        lib = abi.load_library()
        self.c = abi.PtCamera()
        arr = lambda v: (C.c_float * 3)(*_c3(v))  # noqa: E731
        abi.check(lib.pt_camera_init(C.byref(self.c), arr(look_from), arr(look_at), arr(vup), f32(degree_vfov),
                                     f32(aspect_ratio), f32(aperture), f32(focus_dist), f32(time0), f32(time1)),
                  "pt_camera_init")

    def fields(self) -> dict:
        return {n: (list(getattr(self.c, n)) if hasattr(getattr(self.c, n), "__len__") else getattr(self.c, n))
                for n, _ in abi.PtCamera._fields_}


# ---- packing into the C-ABI tables ------------------------------------------------------------------------

class PackedScene:
    """Owns the ctypes tables a PtSceneDesc points into."""

    def __init__(self, hittables, materials, textures, atlas: bytes):
        self.n_hittables, self.n_materials, self.n_textures = len(hittables), len(materials), len(textures)
        self.hittables = (abi.PtHittable * max(1, len(hittables)))(*hittables)
        self.materials = (abi.PtMaterial * max(1, len(materials)))(*materials)
        self.textures = (abi.PtTexture * max(1, len(textures)))(*textures)
        self.atlas = (C.c_uint8 * max(1, len(atlas))).from_buffer_copy(atlas if atlas else b"\0")
        self.atlas_bytes = len(atlas)
        self.desc = abi.PtSceneDesc(self.hittables, self.n_hittables, self.materials, self.n_materials,
                                    self.textures, self.n_textures, 0,
                                    C.cast(self.atlas, C.POINTER(C.c_uint8)), self.atlas_bytes)

    def kinds(self) -> list:
        return [self.hittables[i].kind for i in range(self.n_hittables)]


def pack(hittables: Iterable, atlas: TextureAtlas | None = None) -> PackedScene:
    tex_index: dict = {}
    mat_index: dict = {}
    textures: list = []
    materials: list = []
    used_atlas = atlas

    def tex_id(t) -> int:
        nonlocal used_atlas
        key = t
        if key in tex_index:
            return tex_index[key]
        e = abi.PtTexture()
        if isinstance(t, solid_texture):
            e.kind = abi.PT_TEX_SOLID
            e.color0[:] = t.color
        elif isinstance(t, checker_texture):
            e.kind = abi.PT_TEX_CHECKER
            e.color0[:] = t.odd
            e.color1[:] = t.even
        elif isinstance(t, image_texture):
            e.kind = abi.PT_TEX_IMAGE
            e.width, e.height, e.offset, e.freq = t.width, t.height, t.offset, t.cyclic_frequency
            if t.atlas is not None:
                if used_atlas is None:
                    used_atlas = t.atlas
                elif used_atlas is not t.atlas:
                    raise ValueError("all image textures of a scene must share one atlas")
        else:
            raise TypeError(f"not a texture: {t!r}")
        tex_index[key] = len(textures)
        textures.append(e)
        return tex_index[key]

    def mat_id(m) -> int:
        if m in mat_index:
            return mat_index[m]
        e = abi.PtMaterial()
        e.texture = -1
        if isinstance(m, lambertian_material):
            e.kind, e.texture = abi.PT_MAT_LAMBERTIAN, tex_id(m.albedo)
        elif isinstance(m, metal_material):
            e.kind, e.param = abi.PT_MAT_METAL, m.fuzz
            e.color[:] = m.albedo
        elif isinstance(m, dielectric_material):
            e.kind, e.param = abi.PT_MAT_DIELECTRIC, m.ref_idx
            e.color[:] = m.albedo
        elif isinstance(m, lightsource_material):
            e.kind, e.texture = abi.PT_MAT_LIGHTSOURCE, tex_id(m.emit)
        elif isinstance(m, isotropic_material):
            e.kind, e.texture = abi.PT_MAT_ISOTROPIC, tex_id(m.albedo)
        else:
            raise TypeError(f"not a material: {m!r}")
        mat_index[m] = len(materials)
        materials.append(e)
        return mat_index[m]

    def fill_sphere(f, s: sphere):
        f[0:3] = s.center0
        f[3:6] = s.center1
        f[6], f[7], f[8] = s.radius, s.time0, s.time1

    def fill_box(f, b: box):
        f[0:3] = b.box_min
        f[3:6] = b.box_max

    out = []
    for h in hittables:
        e = abi.PtHittable()
        if isinstance(h, sphere):
            e.kind, e.material = abi.PT_HIT_SPHERE, mat_id(h.material_type)
            fill_sphere(e.f, h)
        elif isinstance(h, _rect):
            e.kind = {xy_rect: abi.PT_HIT_XY_RECT, xz_rect: abi.PT_HIT_XZ_RECT, yz_rect: abi.PT_HIT_YZ_RECT}[type(h)]
            e.material = mat_id(h.material_type)
            e.f[0:5] = (h.a0, h.a1, h.b0, h.b1, h.k)
        elif isinstance(h, triangle):
            e.kind, e.material = abi.PT_HIT_TRIANGLE, mat_id(h.material_type)
            e.strategy = abi.PT_TRI_BADOUEL if h.strategy == "badouel" else abi.PT_TRI_MOLLER_TRUMBORE
            e.f[0:9] = h.v0 + h.v1 + h.v2
        elif isinstance(h, box):
            e.kind, e.material = abi.PT_HIT_BOX, mat_id(h.material_type)
            fill_box(e.f, h)
        elif isinstance(h, constant_medium):
            e.kind, e.material = abi.PT_HIT_CONSTANT_MEDIUM, mat_id(h.phase_function)
            if isinstance(h.boundary, sphere):
                e.boundary_kind = abi.PT_HIT_SPHERE
                fill_sphere(e.f, h.boundary)
            else:
                e.boundary_kind = abi.PT_HIT_BOX
                fill_box(e.f, h.boundary)
            e.f[9] = h.neg_inv_density
        else:
            raise TypeError(f"not a hittable: {h!r}")
        out.append(e)
    atlas_bytes = used_atlas.bytes() if (used_atlas is not None and any(t.kind == abi.PT_TEX_IMAGE for t in textures)) else b""
    return PackedScene(out, materials, textures, atlas_bytes)


def pack_tables(hittables: np.ndarray, materials: Sequence, textures: Sequence, atlas: bytes = b"") -> PackedScene:
    """Bulk path for generated scenes (100k triangles): `hittables` is a structured array with the
    PtHittable layout (see `hittable_dtype`)."""
    hs = (abi.PtHittable * max(1, len(hittables))).from_buffer_copy(np.ascontiguousarray(hittables).tobytes()
                                                                    if len(hittables) else bytes(64))
    ps = PackedScene.__new__(PackedScene)
    ps.n_hittables, ps.n_materials, ps.n_textures = len(hittables), len(materials), len(textures)
    ps.hittables = hs
    ps.materials = (abi.PtMaterial * max(1, len(materials)))(*materials)
    ps.textures = (abi.PtTexture * max(1, len(textures)))(*textures)
    ps.atlas = (C.c_uint8 * max(1, len(atlas))).from_buffer_copy(atlas if atlas else b"\0")
    ps.atlas_bytes = len(atlas)
    ps.desc = abi.PtSceneDesc(ps.hittables, ps.n_hittables, ps.materials, ps.n_materials, ps.textures, ps.n_textures,
                              0, C.cast(ps.atlas, C.POINTER(C.c_uint8)), ps.atlas_bytes)
    return ps


hittable_dtype = np.dtype([("kind", "<i4"), ("material", "<i4"), ("boundary_kind", "<i4"), ("strategy", "<i4"),
                           ("f", "<f4", (12,))])
assert hittable_dtype.itemsize == 64
