"""Cheap guard against the oracle drifting from the reference's text (build container only: skipped where /root/reference
is absent).  Not a parity pin — the reference cannot be built here (every header needs triSYCL) — but it turns "the
restatement was read carefully against the headers" into something that fails loudly: for each literal, comparison form
and ordering the oracle depends on, (a) the reference still says it at the cited place and (b) the oracle says the same."""
import re
from pathlib import Path

import pytest

REF = Path("/root/reference/include")
ROOT = Path(__file__).resolve().parent.parent
ORACLE = (ROOT / "oracle" / "pt_oracle.c").read_text()
HEADER = (ROOT / "include" / "pt_render.h").read_text()

pytestmark = pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")


def norm(s: str) -> str:
    return re.sub(r"\s+", " ", s)


# (reference file, regex over its whitespace-normalised text, regex over the oracle's whitespace-normalised text, what)
CHECKS = [
    ("render.hpp", r"arg\.hit\(ctx, r, 0\.001f, closest_so_far", r"0\.001f, closest_so_far", "hit_world: tmin = 0.001, shrinking tmax"),
    ("render.hpp", r"auto closest_so_far = infinity", r"float closest_so_far = PT_INF", "hit_world starts at +inf"),
    ("render.hpp", r"auto constexpr depth = 50", r"", "depth 50 (the callers' default)"),
    ("render.hpp", r"0\.5f \* \(unit_direction\.y\(\) \+ 1\.0f\)", r"0\.5f \* \(\w+\.y \+ 1\.0f\)", "sky blend factor"),
    ("render.hpp", r"color \{ 0\.5f, 0\.7f, 1\.0f \}", r"V\(0\.5f, 0\.7f, 1\.0f\)", "sky colour"),
    ("render.hpp", r"color cur_attenuation \{ 1\.0f, 1\.0f, 1\.0f \}", r"V\(1\.0f, 1\.0f, 1\.0f\)", "attenuation starts at 1"),
    ("sphere.hpp", r"if \(temp < max && temp > min\)", r"if \(temp < mx && temp > mn\)", "sphere roots: strict on both ends"),
    ("sphere.hpp", r"if \(discriminant > 0\)", r"if \(discriminant > 0\)", "sphere: discriminant strictly positive"),
    ("rectangle.hpp", r"if \(t < min \|\| t > max\)", r"if \(t < mn \|\| t > mx\)", "rect: inclusive t range, NaN passes"),
    ("triangle.hpp", r"constexpr auto epsilon = 0\.0000001f", r"const float epsilon = 0\.0000001f", "Moller-Trumbore epsilon"),
    ("triangle.hpp", r"if \(a_abs < epsilon\)", r"if \(a_abs < epsilon\)", "triangle: parallel rejection"),
    ("triangle.hpp", r"if \(length < min \|\| length > max\)", r"if \(length < mn \|\| length > mx\)", "triangle: inclusive t range"),
    ("constant_medium.hpp", r"arg\.hit\(ctx, r, -infinity, infinity, rec1", r"boundary_hit\(c, h, r, -PT_INF, PT_INF, &rec1\)", "medium: first boundary hit over (-inf, inf)"),
    ("constant_medium.hpp", r"rec1\.t \+ 0\.0001f, infinity, rec2", r"rec1\.t \+ 0\.0001f, PT_INF, &rec2", "medium: second boundary hit from t1 + 1e-4"),
    ("constant_medium.hpp", r"if \(rec1\.t < min\)", r"if \(rec1\.t < mn\) rec1\.t = mn", "medium: clamp to tmin"),
    ("constant_medium.hpp", r"if \(rec2\.t > max\)", r"if \(rec2\.t > mx\) rec2\.t = mx", "medium: clamp to tmax"),
    ("rtweekend.hpp", r"auto z = \(float_t\(\) > 0\.5\) \? absz : -absz", r"> 0\.5f\) \? absz : -absz", "unit_vec: z sign from the third draw"),
    ("rtweekend.hpp", r"auto theta = float_t\(0, 2 \* pi\); auto phi = float_t\(0, pi\)", r"", "in_unit_ball: r, theta, phi draw order"),
    ("material.hpp", r"sycl::pow\(\(1 - cosine\), 5\.0f\)", r"m_pow5\(1\.0f - cosine\)", "Schlick: fifth power"),
    ("xorshift.hpp", r"state \^= state >> 7; state \^= state << 1; state \^= state >> 9", r">> 7.{0,40}<< 1.{0,40}>> 9", "xorshift32 triple (7, 1, 9)"),
    ("texture.hpp", r"std::vector<uint8_t> image_texture::texture_data \{ 0, 0, 1 \}", r"", "atlas starts with the {0,0,1} fallback texel"),
]


@pytest.mark.parametrize("file,ref_rx,orc_rx,what", CHECKS, ids=[c[3] for c in CHECKS])
def test_reference_text_and_oracle_agree(file, ref_rx, orc_rx, what):
    ref = norm((REF / file).read_text())
    assert re.search(ref_rx, ref), f"the reference no longer says this in {file}: {what}"
    if orc_rx:
        assert re.search(orc_rx, norm(ORACLE)), f"the oracle no longer says this: {what}"


def test_box_side_order_and_variant_orders():
    """box.hpp:20-25 side order (xy@z1, xy@z0, xz@y1, xz@y0, yz@x1, yz@x0) and the std::variant alternative orders that
    define the ABI tags (render.hpp:22-23, material.hpp:133-135, texture.hpp:154, rectangle.hpp:130, constant_medium.hpp:10)."""
    box = norm((REF / "box.hpp").read_text())
    sides = re.findall(r"sides\[(\d)\] = (xy|xz|yz)_rect\(([^;]*?), mat_type\)", box)
    assert [(int(i), k, a.replace(" ", "")) for i, k, a in sides] == [
        (0, "xy", "p0.x(),p1.x(),p0.y(),p1.y(),p1.z()"), (1, "xy", "p0.x(),p1.x(),p0.y(),p1.y(),p0.z()"),
        (2, "xz", "p0.x(),p1.x(),p0.z(),p1.z(),p1.y()"), (3, "xz", "p0.x(),p1.x(),p0.z(),p1.z(),p0.y()"),
        (4, "yz", "p0.y(),p1.y(),p0.z(),p1.z(),p1.x()"), (5, "yz", "p0.y(),p1.y(),p0.z(),p1.z(),p0.x()")]
    # the oracle's box_hit switch lists the same six calls in the same order
    orc_sides = re.findall(r"rect_hit\(c, (\d), (\w\d), (\w\d), (\w\d), (\w\d), (\w\d), r, mn, closest_so_far", norm(ORACLE))
    assert orc_sides == [("0", "x0", "x1", "y0", "y1", "z1"), ("0", "x0", "x1", "y0", "y1", "z0"),
                         ("1", "x0", "x1", "z0", "z1", "y1"), ("1", "x0", "x1", "z0", "z1", "y0"),
                         ("2", "y0", "y1", "z0", "z1", "x1"), ("2", "y0", "y1", "z0", "z1", "x0")]
    variants = {
        "render.hpp": (r"std::variant<sphere, xy_rect, triangle, box, constant_medium>",
                       r"PT_HIT_SPHERE = 0, PT_HIT_XY_RECT = 1, PT_HIT_TRIANGLE = 2, PT_HIT_BOX = 3, PT_HIT_CONSTANT_MEDIUM = 4"),
        "material.hpp": (r"std::variant<lambertian_material, metal_material, dielectric_material, lightsource_material, isotropic_material>",
                         r"PT_MAT_LAMBERTIAN = 0, PT_MAT_METAL = 1, PT_MAT_DIELECTRIC = 2, PT_MAT_LIGHTSOURCE = 3, PT_MAT_ISOTROPIC = 4"),
        "texture.hpp": (r"std::variant<checker_texture, solid_texture, image_texture>",
                        r"PT_TEX_CHECKER = 0, PT_TEX_SOLID = 1, PT_TEX_IMAGE = 2"),
        "rectangle.hpp": (r"std::variant<xy_rect, xz_rect, yz_rect>", None),
        "constant_medium.hpp": (r"std::variant<sphere, box>", None),
    }
    header = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)  # tag enums without their comments
    for file, (ref_rx, abi_rx) in variants.items():
        assert re.search(ref_rx, norm((REF / file).read_text())), file
        if abi_rx:
            assert re.search(abi_rx.replace(", ", r",\s*"), norm(header)), f"include/pt_render.h tags no longer follow {file}"
