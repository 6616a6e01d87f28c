"""The C++20 host facade (path_tracer_amd/include/pt/path_tracer.hpp): scenes built with the reference-shaped
C++ constructors flatten to the same C-ABI tables as the Python mirror (CPU), and render to the same
framebuffer through pt_render_host (GPU)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import scenes_small as S
from conftest import assert_bit_identical
from path_tracer_amd import abi, scenes
from path_tracer_amd.scene import (box, checker_texture, constant_medium, dielectric_material, lambertian_material,
                                   lightsource_material, metal_material, pack, sphere, triangle, xy_rect, xz_rect,
                                   yz_rect, camera)

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "cpp" / "facade_main.cpp"


@pytest.fixture(scope="module")
def facade_bin(tmp_path_factory, lib):
    out = tmp_path_factory.mktemp("facade") / "facade_main"
    libdir = ROOT / "path_tracer_amd"
    subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", f"-I{ROOT / 'path_tracer_amd' / 'include'}",
                    str(SRC), "-o", str(out), f"-L{libdir}", "-lpt_render", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def zoo_py():
    checker = checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    hs = [
        sphere((0, -100.5, -1), 100, lambertian_material(checker)),
        sphere((0, 0, -1), 0.5, lambertian_material((0.7, 0.3, 0.3))),
        sphere((1, 0, -1), 0.5, metal_material((0.8, 0.6, 0.2), 0.3)),
        sphere((-1, 0, -1), 0.5, dielectric_material(1.5, (1, 1, 1))),
        sphere((0.3, 0.1, -0.2), (0.3, 0.3, -0.2), 0.0, 1.0, 0.12, lambertian_material((0.7, 0.3, 0.3))),
        triangle((-0.5, 0.6, -1.2), (0.5, 0.6, -1.2), (0, 1.3, -0.9), lambertian_material((0.1, 0.2, 0.9))),
        box((1.2, -0.5, -2.5), (1.8, 0.7, -1.9), metal_material((0.7, 0.6, 0.5), 7.0)),
        constant_medium(sphere((0.8, 0.9, -1.2), 0.4, lambertian_material((1, 1, 1))), 3.0, (0.9, 0.9, 1.0)),
        xz_rect(-1, 1, -2, 0, 2.5, lightsource_material((4, 4, 4))),
        yz_rect(-0.5, 1.5, -2.5, -0.5, -2.2, lambertian_material((0.2, 0.8, 0.2))),
        xy_rect(-2, -1, -0.5, 1, -1.5, lambertian_material((0.7, 0.3, 0.3))),
        constant_medium(box((-1.9, -0.5, -0.9), (-1.3, 0.2, -0.3), lambertian_material((1, 1, 1))), 5.0, checker),
    ]
    cam = dict(look_from=(0.3, 0.6, 2.5), look_at=(0, 0.2, -1), vup=(0, 1, 0), vfov=50.0, aperture=0.1,
               focus_dist=3.4, time0=0.0, time1=1.0)
    return pack(hs), cam


def read_dump(path):
    raw = Path(path).read_bytes()
    n = np.frombuffer(raw[:12], dtype=np.int32)
    off = 12
    sizes = (n[0] * 64, n[1] * 32, n[2] * 48, 96)
    parts = []
    for s in sizes:
        parts.append(raw[off:off + s])
        off += s
    assert off == len(raw)
    return n.tolist(), parts


@pytest.mark.parametrize("name", ["cornell", "zoo"])
def test_cpp_tables_match_python_packer(facade_bin, tmp_path, name):
    out = tmp_path / f"{name}.bin"
    subprocess.run([str(facade_bin), "dump", name, str(out)], check=True)
    n, (hb, mb, tb, cb) = read_dump(out)
    if name == "cornell":
        ps, cam = S.cornell_scene()
        c = camera(cam["look_from"], cam["look_at"], cam["vup"], cam["vfov"], np.float32(64) / np.float32(36),
                   cam["aperture"], cam["focus_dist"], cam["time0"], cam["time1"])
    else:
        ps, cam = zoo_py()
        c = camera(cam["look_from"], cam["look_at"], cam["vup"], cam["vfov"], 1.5, cam["aperture"], cam["focus_dist"],
                   cam["time0"], cam["time1"])
    assert n == [ps.n_hittables, ps.n_materials, ps.n_textures]
    assert hb == bytes(ps.hittables)[:len(hb)]
    assert mb == bytes(ps.materials)[:len(mb)]
    assert tb == bytes(ps.textures)[:len(tb)]
    assert cb == bytes(c.c)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "zoo"])
def test_cpp_render_matches_python_render(facade_bin, tmp_path, orc, name):
    from path_tracer_amd import render as R
    w, h, spp = 64, 40, 6
    out = tmp_path / f"{name}.f32"
    subprocess.run([str(facade_bin), "render", name, str(w), str(h), str(spp), str(out)], check=True)
    fb = np.fromfile(out, dtype=np.float32).reshape(h, w, 3)
    ps, cam = S.cornell_scene() if name == "cornell" else zoo_py()
    c = scenes.make_camera(cam, w, h)
    assert_bit_identical(fb, R.render_host(w, h, spp, ps, c), f"C++ facade vs Python host: {name}")
    orc.set_math(True)
    assert_bit_identical(fb, orc.render(ps, c.c, w, h, spp), f"C++ facade vs oracle: {name}")


@pytest.fixture(scope="module")
def example_bin(tmp_path_factory, lib):
    out = tmp_path_factory.mktemp("example") / "sycl-rt-mi355x"
    libdir = ROOT / "path_tracer_amd"
    subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", f"-I{ROOT / 'path_tracer_amd' / 'include'}",
                    str(ROOT / "examples" / "smoke_sphere.cpp"), "-o", str(out), f"-L{libdir}", "-lpt_render",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


@pytest.fixture(scope="module")
def images_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("images")
    scenes.export_reference_textures(d)  # Xilinx.ppm / SYCL.ppm from the committed decoded-pixel fixture
    return d


@pytest.mark.parametrize("textures", ["reference", "procedural"])
def test_example_main_builds_the_same_scene_as_python(example_bin, tmp_path, images_dir, textures):
    """examples/smoke_sphere.cpp (the reference's main.cpp against the C++ facade) and scenes.smoke_sphere_scene (Python)
    construct the 496-hittable default scene with the same RNG draw order and binary32 arithmetic — with the reference's
    two images loaded by the C++ host's own image_texture_factory (PPM) or with the generated stand-ins."""
    out = tmp_path / "smoke.bin"
    subprocess.run([str(example_bin), "400", "225", "1", str(tmp_path / "x.ppm"), str(out),
                    str(images_dir) if textures == "reference" else "procedural"], check=True)
    n, (hb, mb, tb, cb) = read_dump(out)
    ps, cam = scenes.build("smoke", textures=textures)
    c = scenes.make_camera(cam, 400, 225)
    assert n == [ps.n_hittables, ps.n_materials, ps.n_textures]
    assert hb == bytes(ps.hittables)[:len(hb)]
    assert mb == bytes(ps.materials)[:len(mb)]
    assert tb == bytes(ps.textures)[:len(tb)]
    assert cb == bytes(c.c)
    if textures == "reference":  # texture.hpp:113-114: fallback texel, Xilinx.jpg at texel 1, SYCL.png right behind it
        assert len(ps.atlas) == 3 + 1024 * 512 * 3 + 1280 * 559 * 3
        tex = np.frombuffer(tb, dtype=np.uint8).reshape(n[2], 48)
        img = tex[tex[:, :4].copy().view(np.int32)[:, 0] == abi.PT_TEX_IMAGE]
        assert img[:, 28:40].copy().view(np.uint32).tolist() == [[1024, 512, 1], [1280, 559, 1 + 1024 * 512]]


def test_example_main_missing_image_gets_the_fallback_texel(example_bin, tmp_path):
    """texture.hpp:106-111: a load failure is a message on stderr and the 1x1 texture at offset 0 — never an error exit."""
    (tmp_path / "imgs").mkdir()
    (tmp_path / "imgs" / "SYCL.ppm").write_bytes(b"P3\n1 1\n255\n0 0 0\n")  # a format this host does not decode
    out = tmp_path / "smoke.bin"
    r = subprocess.run([str(example_bin), "40", "22", "1", str(tmp_path / "x.ppm"), str(out), str(tmp_path / "imgs")],
                       check=True, capture_output=True, text=True)
    assert r.stderr.count("ERROR: Could not load texture image file") == 2
    assert "can't fopen" in r.stderr and "unknown image type" in r.stderr
    n, (hb, mb, tb, cb) = read_dump(out)
    ps, _ = scenes.build("smoke", textures="procedural")
    assert n[0] == ps.n_hittables
    tex = np.frombuffer(tb, dtype=np.uint8).reshape(n[2], 48)
    kinds = tex[:, :4].copy().view(np.int32)[:, 0]
    img = tex[kinds == abi.PT_TEX_IMAGE]
    assert len(img) == 2
    whoff = img[:, 28:40].copy().view(np.uint32)  # PtTexture: kind, color0[3], color1[3], width, height, offset, freq
    assert whoff.tolist() == [[1, 1, 0], [1, 1, 0]]
    assert img[:, 40:44].copy().view(np.float32)[:, 0].tolist() == [1.0, 5.0]


@pytest.mark.skipif(not Path("/root/reference/images/Xilinx.jpg").exists(), reason="the reference tree is only present in the build container")
def test_example_main_opens_the_references_own_image_files(example_bin, tmp_path):
    """run with the reference's own ../images directory, the C++ host decodes Xilinx.jpg and SYCL.png itself
    (pt/image_io.hpp) and builds byte for byte the tables and the atlas layout the Python host builds from the same files (PIL)."""
    out = tmp_path / "smoke.bin"
    r = subprocess.run([str(example_bin), "400", "225", "1", str(tmp_path / "x.png"), str(out), "/root/reference/images"], check=True,
                       capture_output=True, text=True)
    assert "ERROR" not in r.stderr
    n, (hb, mb, tb, cb) = read_dump(out)
    ps, cam = scenes.build("smoke")
    assert n == [ps.n_hittables, ps.n_materials, ps.n_textures]
    assert hb == bytes(ps.hittables)[:len(hb)] and mb == bytes(ps.materials)[:len(mb)] and tb == bytes(ps.textures)[:len(tb)]


@pytest.mark.gpu
def test_example_main_renders_the_python_frame(example_bin, tmp_path, images_dir, orc):
    from path_tracer_amd import png
    from path_tracer_amd import render as R
    # the decoded-pixel fixture as the files a main.cpp-style caller finds: SYCL.png (lossless: written from the fixture, decoded by the
    # C++ host's own PNG decoder) beside Xilinx.ppm (a JPEG cannot be regenerated bit for bit; the .jpg path is covered on the CPU)
    import shutil
    xil, syc = scenes.reference_textures()
    shutil.copy(images_dir / "Xilinx.ppm", tmp_path / "Xilinx.ppm")
    png.write_png(str(tmp_path / "SYCL.png"), syc)
    images_dir = tmp_path
    ps, cam = scenes.build("smoke")
    fb = R.render_host(96, 54, 8, ps, scenes.make_camera(cam, 96, 54))
    want = orc.tonemap_rgb8(fb)
    ppm = tmp_path / "out.ppm"
    subprocess.run([str(example_bin), "96", "54", "8", str(ppm), "-", str(images_dir)], check=True)
    raw = ppm.read_bytes()
    header, body = raw.split(b"\n255\n", 1)
    assert header.startswith(b"P6\n96 54")
    np.testing.assert_array_equal(np.frombuffer(body, dtype=np.uint8).reshape(54, 96, 3), want)
    # ... and out.png (main.cpp:57), written by the C++ host's own PNG writer
    out_png = tmp_path / "out.png"
    subprocess.run([str(example_bin), "96", "54", "8", str(out_png), "-", str(images_dir)], check=True)
    from PIL import Image
    np.testing.assert_array_equal(np.asarray(Image.open(out_png).convert("RGB")), want)


# ---- the N-GPU host path in C++ (pt/distributed.hpp + libpt_dist.so: RCCL gather + un-interleave, no Python) ------------

@pytest.fixture(scope="module")
def dist_bin(tmp_path_factory, lib):
    out = tmp_path_factory.mktemp("dist") / "dist_main"
    libdir = ROOT / "path_tracer_amd"
    assert (libdir / "libpt_dist.so").exists(), "libpt_dist.so is built by __graft_entry__.build()"
    subprocess.run(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    f"-I{libdir / 'include'}", str(ROOT / "tests" / "cpp" / "dist_main.cpp"), "-o", str(out), f"-L{libdir}",
                    "-lpt_dist", "-lpt_render", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def test_dist_library_exports_its_header(lib):
    """include/pt_dist.h <-> libpt_dist.so (no compute: loadable without a GPU as long as RCCL's library is present)."""
    import ctypes
    import re
    hdr = (ROOT / "include" / "pt_dist.h").read_text()
    names = sorted(set(re.findall(r"\b(pt_dist_[a-z_]+)\s*\(", hdr)))
    assert names == ["pt_dist_gather_floats", "pt_dist_gather_frame", "pt_dist_last_error", "pt_dist_render"]
    d = ctypes.CDLL(str(ROOT / "path_tracer_amd" / "libpt_dist.so"))
    for n in names:
        assert hasattr(d, n), n
    d.pt_dist_gather_floats.restype = ctypes.c_int64
    p = abi.PtRenderParams(21, 13, 1, 50, 0, 4, 0, 0)  # 3x2 tiles -> 2 per shard x 4 shards x 64 pixels x 3
    assert d.pt_dist_gather_floats(ctypes.byref(p)) == 4 * 2 * 64 * 3


@pytest.mark.gpu
@pytest.mark.parametrize("mode,n", [("multi", 1), ("shards", 1), ("shards", 3), ("shards", 8)])
def test_cpp_multi_gpu_host_path(dist_bin, tmp_path, orc, mode, n):
    """render_multi_gpu (ncclCommInitAll + thread per GPU + pt_dist_render; one device on this box) and the shard replay
    (n ranks' pt_render outputs placed where ncclGather puts them, then pt_unshard_tiles) give the oracle's frame."""
    w, h, spp = 70, 42, 5
    out = tmp_path / "frame.f32"
    subprocess.run([str(dist_bin), mode, str(w), str(h), str(spp), str(out), str(n)], check=True, timeout=300)
    fb = np.fromfile(out, dtype=np.float32).reshape(h, w, 3)
    ps, cam = S.cornell_scene()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    assert_bit_identical(fb, orc.render(ps, c.c, w, h, spp), f"C++ {mode} n={n}")
