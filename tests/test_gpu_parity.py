"""GPU parity tests (run with -m gpu on an MI355X): the hand-written HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT float32 (any NaN == any NaN) against the oracle in portable-math mode — the project's
pinned definition of the seven transcendentals — for every function-level probe and every framebuffer;
against the oracle in glibc mode (the reference's own libm semantics on this host) the comparison is
statistical because one differing ulp re-rolls a pixel's RNG stream: 8-bit PSNR >= 38 dB
at 96x54x16 spp and mean radiance within 2 %.  Scenes without transcendentals (Cornell) are bit-exact in
both modes."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import scenes_small as S
from conftest import assert_bit_identical, bits, psnr_8bit
from dist_util import unshard_reference
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"
FP = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def gpu_math(lib, op, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    out = np.empty_like(a)
    abi.check(lib.pt_debug_math(op, a.ctypes.data_as(FP), bb.ctypes.data_as(FP) if bb is not None else None,
                                out.ctypes.data_as(FP), a.size), "pt_debug_math")
    return out


def test_native_extension_is_what_runs(lib, torch_gpu):
    assert abi.library_path().exists()
    assert lib.pt_abi_version() == abi.PT_ABI_VERSION
    out = gpu_math(lib, 7, np.float32([4.0, 2.0]))
    assert out.tolist() == [2.0, float(np.sqrt(np.float32(2.0)))]


@pytest.mark.parametrize("op", range(7))
def test_math_bit_exact(lib, orc, op):
    rng = np.random.default_rng(100 + op)
    n = 300_000
    u = rng.random(n, dtype=np.float32)
    special = np.float32([0, -0.0, 1, -1, 0.5, np.inf, -np.inf, np.nan, 1e-30, 1e-40, 3e38, 1e9, 2e9, -5e8, 8388608.5,
                          0.99999994, 2.0 ** -32, 1.5, -2.5])
    if op in (0, 1):
        a = np.concatenate([u * np.float32(2 * np.pi), (u - 0.5) * 1e5, (u - 0.5) * 1e-2, (u - 0.5) * 4e9, special])
    elif op == 2:
        a = np.concatenate([u, u * 1e-5, u * 1e-38, u * 100, special])
    elif op == 3:
        a = np.concatenate([u * 2, u * 1e-7, special])
    elif op == 5:
        a = np.concatenate([u * 2 - 1, (u - 0.5) * 1e-4, special])
    else:
        a = np.concatenate([(u - 0.5) * 40, (u - 0.5) * 1e-3, special])
    a = a.astype(np.float32)
    b = None
    if op == 4:
        b = np.concatenate([(rng.random(len(a) - len(special), dtype=np.float32) - 0.5) * 40, special[::-1]]).astype(np.float32)
    orc.set_math(True)
    assert_bit_identical(gpu_math(lib, op, a, b), orc.math(op, a, b), f"math op {op}")
    if op in (0, 1):  # the fused sin / cos of the unit-ball sampler (pt_math.hpp: sincosf_): the same bits as the two functions
        assert_bit_identical(gpu_math(lib, 14 + op, a, None), orc.math(op, a, None), f"fused sincos, output {op}")


def test_ieee_sqrt_and_div(lib):
    """The kernel relies on correctly rounded fp32 sqrt and division, denormals included."""
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.random(400_000, dtype=np.float32) * 1e6, rng.random(1000, dtype=np.float32) * 1e-38,
                        np.float32([0, 1e-45, 3e38, np.inf, 2, 3])]).astype(np.float32)
    assert_bit_identical(gpu_math(lib, 7, a), np.sqrt(a), "sqrt")
    x = ((rng.random(len(a), dtype=np.float32) - 0.5) * np.float32(10) ** rng.integers(-30, 30, len(a))).astype(np.float32)
    y = ((rng.random(len(a), dtype=np.float32) - 0.5) * np.float32(10) ** rng.integers(-30, 30, len(a))).astype(np.float32)
    x[:6] = [0, 1, -1, 1e-40, 3e38, np.inf]
    y[:6] = [0, 0, 1e-40, 3, 1e-38, np.inf]
    with np.errstate(all="ignore"):
        assert_bit_identical(gpu_math(lib, 8, x, y), x / y, "div")


def test_fast_division_is_exact(lib):
    """The shared-reciprocal quotient used for rect/box sides (pt_device.hpp: div_exact) equals the IEEE quotient
    bit for bit over its guarded range: |d| in [2^-40, 2^40], n = 0 or |n| in [2^-100, 2^61], incl. all-ones /
    all-zeros significands."""
    rng = np.random.default_rng(77)
    n_ = 4_000_000
    md = rng.integers(0, 2 ** 23, n_, dtype=np.uint32)
    mn = rng.integers(0, 2 ** 23, n_, dtype=np.uint32)
    md[::5] = 0x7FFFFF - rng.integers(0, 256, len(md[::5]), dtype=np.uint32)
    md[1::5] = rng.integers(0, 256, len(md[1::5]), dtype=np.uint32)
    mn[2::7] = 0x7FFFFF - rng.integers(0, 4, len(mn[2::7]), dtype=np.uint32)
    ed = rng.integers(-40, 41, n_).astype(np.uint32) + 127
    en = rng.integers(-100, 62, n_).astype(np.uint32) + 127
    sd = rng.integers(0, 2, n_, dtype=np.uint32) << 31
    sn = rng.integers(0, 2, n_, dtype=np.uint32) << 31
    d = ((ed << 23) | md | sd).astype(np.uint32).view(np.float32)
    n = ((en << 23) | mn | sn).astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        assert_bit_identical(gpu_math(lib, 9, n, d), n / d, "shared-reciprocal division")
    # n = +-0: the quotient is a zero whose sign may differ from IEEE's; any |t| < 2^-60 is below the rect
    # test's min = 0.001 and rejected either way (pt_device.hpp: rect_fast)
    z = np.zeros(2000, np.float32)
    z[1000:] = -0.0
    assert (gpu_math(lib, 9, z, d[:2000]) == 0).all()


def test_guarded_reciprocal_is_correctly_rounded(lib):
    """The ray context's RN(1/d) = hardware estimate + one Newton step in fma arithmetic (pt_device.hpp: rcp_rn_guarded):
    equal to the IEEE quotient for EVERY significand, both signs, exponents across the guarded range [2^-40, 2^40]
    (the computation is scale-invariant there)."""
    m = np.arange(2 ** 23, dtype=np.uint32)
    for e in (-40, -7, -1, 0, 1, 2, 23, 40):
        for sgn in (0, 1):
            d = ((np.uint32(e + 127) << 23) | m | (np.uint32(sgn) << 31)).astype(np.uint32).view(np.float32)
            with np.errstate(all="ignore"):
                assert_bit_identical(gpu_math(lib, 10, d, None), (np.float32(1.0) / d).astype(np.float32), f"1/d, exponent {e}")


def test_unit_range_sqrt_is_correctly_rounded(lib):
    """sqrt_rn_unit (pt_device.hpp: v_sqrt_f32 + the two-sided neighbour test, no rescaling): the IEEE square root for
    EVERY float in [2^-60, 4) and for 0 — the only values 1 - x*x and maxy*maxy - y*y of rtweekend.hpp:60-67,83-88 take
    (multiples of 2^-48 in [0, 1])."""
    m = np.arange(2 ** 23, dtype=np.uint32)
    for e in range(-60, 2):
        x = ((np.uint32(e + 127) << 23) | m).astype(np.uint32).view(np.float32)
        assert_bit_identical(gpu_math(lib, 11, x, None), np.sqrt(x), f"sqrt, exponent {e}")
    z = np.float32([0.0, 1.0, 2.0 ** -48, 3.9999998])
    assert_bit_identical(gpu_math(lib, 11, z, None), np.sqrt(z), "sqrt edge values")


def test_sky_unit_direction_shortcut_is_exact(lib):
    """render.hpp:83-85 for a regular ray (pt_device.hpp: sky_unit_y): unit_vector(d).y = d.y / sqrt(d.d) through the hardware square
    root + neighbour test and ONE correctly rounded reciprocal + correction instead of the IEEE expansions — the same bits for
    directions over the whole guarded range 2^-40 <= |d_c| <= 2^40 (d.d from 3 * 2^-80 to 3 * 2^80), adversarial significands
    included; and the square root form itself for EVERY float of four binades outside sqrt_rn_unit's own test range (it is
    scale-invariant under x -> 4 x: an even and an odd exponent cover all)."""
    rng = np.random.default_rng(41)
    n_ = 3_000_000

    def comp(exp_lo, exp_hi):
        m = rng.integers(0, 2 ** 23, n_, dtype=np.uint32)
        m[::7] = 0x7FFFFF - rng.integers(0, 8, len(m[::7]), dtype=np.uint32)
        m[3::7] = rng.integers(0, 8, len(m[3::7]), dtype=np.uint32)
        e = rng.integers(exp_lo, exp_hi + 1, n_).astype(np.uint32) + 127
        sgn = rng.integers(0, 2, n_, dtype=np.uint32) << 31
        return ((e << 23) | m | sgn).astype(np.uint32).view(np.float32)

    for lo, hi in ((-40, 40), (-2, 2), (38, 40), (-40, -38)):
        dx, dy, dz = comp(lo, hi), comp(lo, hi), comp(lo, hi)
        a = ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)  # dot(d, d), left to right in binary32
        want = (dy / np.sqrt(a)).astype(np.float32)
        assert_bit_identical(gpu_math(lib, 12, dy, a), want, f"d.y / sqrt(d.d), exponents {lo}..{hi}")
    m = np.arange(2 ** 23, dtype=np.uint32)
    for e in (-81, -80, 80, 81):
        x = ((np.uint32(e + 127) << 23) | m).astype(np.uint32).view(np.float32)
        assert_bit_identical(gpu_math(lib, 11, x, None), np.sqrt(x), f"sqrt, exponent {e}")


def test_checker_sign_shortcut_is_exact(lib, orc):
    """texture.hpp:43-45 (pt_device.hpp: checker_sines_negative): `sin(a) sin(b) sin(c) < 0` decided from the range reductions alone for
    regular arguments, from the product itself otherwise — against the oracle's product on random arguments of every magnitude, on the
    neighbours of multiples of pi/2 (where a sine changes sign or is smallest), and on zeros, denormals, huge values, infinities and NaNs
    (where the product underflows, is a zero or a NaN).  tests/cpp/checker_sign_exhaustive.c covers every regular float on the CPU."""
    rng = np.random.default_rng(77)
    n = 2_000_000
    e = rng.integers(-40, 36, n).astype(np.uint32) + 127
    a = ((e << 23) | rng.integers(0, 2 ** 23, n, dtype=np.uint32) | (rng.integers(0, 2, n, dtype=np.uint32) << 31)).view(np.float32)
    k = rng.integers(-200000, 200000, 400_000)
    near = (k * (np.pi / 2)).astype(np.float32)
    near = (near.view(np.int32) + rng.integers(-3, 4, len(near)).astype(np.int32)).view(np.float32)
    special = np.float32([0, -0.0, 1e-45, -1e-45, 1e-38, 2.0 ** -30, np.nextafter(np.float32(2.0 ** -30), np.float32(0)), -(2.0 ** -30),
                          2.0 ** 30, np.nextafter(np.float32(2.0 ** 30), np.float32(0)), -(2.0 ** 30), 3e38, np.inf, -np.inf, np.nan,
                          1e-20, -1e-20, 1e-15, 3.1415927, -3.1415927, 6.2831855, 1.5707964, 10.0, -10.0])
    a = np.concatenate([a, near, special, (rng.random(500_000, dtype=np.float32) - 0.5) * np.float32(2e4)]).astype(np.float32)
    b = rng.permutation(a).astype(np.float32)
    b[-2000:] = rng.choice(special, 2000)
    a[:2000] = rng.choice(special, 2000)
    orc.set_math(True)
    sa, sb = orc.math(0, a, None), orc.math(0, b, None)
    s1 = orc.math(0, np.float32([1.0]), None)[0]
    with np.errstate(all="ignore"):
        want = (((sa * sb).astype(np.float32) * s1).astype(np.float32) < 0).astype(np.float32)
    got = gpu_math(lib, 13, a, b)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} differ, first a={a[bad[0]]!r} b={b[bad[0]]!r} got {got[bad[0]]} want {want[bad[0]]}"
    assert want.sum() > 100_000 and (1 - want).sum() > 100_000


def _unit_normals(rng, n, kind):
    v = rng.normal(size=(n, 3))
    if kind == 1: v[:, rng.integers(0, 3)] *= 1e-3       # near a coordinate plane: the seam, the equator
    if kind == 2: v[:, 1] *= 30                          # near the poles
    if kind == 3: v[:, 0] *= 1e-5; v[:, 2] *= 1e-5       # at the poles
    if kind == 4: v[:, 2] = np.abs(v[:, 2]) * 1e-6; v[:, 0] = -np.abs(v[:, 0])  # the seam phi = +-pi
    v /= np.linalg.norm(v, axis=1)[:, None]
    return (v * (1 + rng.normal(size=(n, 1)) * 1e-7)).astype(np.float32)  # (p - centre) / radius is a unit vector up to rounding


@pytest.mark.parametrize("freq,w,h", [(1.0, 1024, 512), (5.0, 320, 140), (0.37, 7, 3), (64.0, 4096, 4096), (1.0, 65536, 65536)])
def test_sphere_texel_fast_path_is_exact(lib, orc, freq, w, h):
    """texture.hpp:140-157 over sphere.hpp:13-24 (pt_device.hpp: sphere_texel_fast): the texel an image texture selects on a sphere, from
    binary32 approximations of atan2 / asin where their error cannot change a floor(), from the reference's chain otherwise.  What the
    kernels take must equal the chain on every normal; the chain itself is pinned to the oracle's math here; and the short form must be
    the one that decides nearly always (it is the point of having it) at the reference's texture sizes."""
    rng = np.random.default_rng(int(w * 31 + h))
    n = 4_000_000
    nn = np.concatenate([_unit_normals(rng, n // 5, k) for k in range(5)])
    special = np.float32([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [0, 0, 0], [np.nan, 0, 1], [0.6, np.nan, 0.8],
                          [0, 1.0000001, 0], [1e-30, 1, 1e-30], [-1, 0, -0.0], [-1, 0, 1e-45], [np.inf, 0, 1], [0.5, 0.5, 0.70710677]])
    nn = np.ascontiguousarray(np.concatenate([nn, special]).astype(np.float32))
    m = len(nn)
    out, exact, fast = np.zeros((m, 2), np.int32), np.zeros((m, 2), np.int32), np.zeros(m, np.uint8)
    uv4 = np.zeros((m, 4), np.float32)
    abi.check(lib.pt_debug_sphere_texel(nn.ctypes.data_as(abi._FP), m, float(freq), w, h, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                        exact.ctypes.data_as(C.POINTER(C.c_int32)), fast.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        uv4.ctypes.data_as(abi._FP)), "pt_debug_sphere_texel")
    # the deviation of the binary32 approximations from the reference's chain, measured on the device: the short form assumes 1e-6
    valid = np.isfinite(uv4).all(axis=1) & (np.abs(nn[:, 1]) <= 1) & (np.maximum(np.abs(nn[:, 0]), np.abs(nn[:, 2])) > 0)
    du = np.abs(uv4[valid, 2].astype(np.float64) - uv4[valid, 0])
    du = np.minimum(du, 1.0 - du)  # (the seam: u = 0 and u = 1 are the same direction)
    dv = np.abs(uv4[valid, 3].astype(np.float64) - uv4[valid, 1])
    assert valid.sum() > 3_500_000 and du.max() <= 2.5e-7 and dv.max() <= 2.5e-7, (du.max(), dv.max())  # (measured: 1.2e-7 both)
    bad = np.nonzero((out != exact).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} texels differ, first normal {nn[bad[0]]!r}: took {out[bad[0]]} chain {exact[bad[0]]} fast={fast[bad[0]]}"
    # the chain, restated with the oracle's math (texture.hpp:140-157, sphere.hpp:13-24; binary32 left to right)
    orc.set_math(True)
    f32, PI = np.float32, np.float32(3.14159265358979323846)
    with np.errstate(all="ignore"):
        phi, theta = orc.math(4, nn[:, 2].copy(), nn[:, 0].copy()), orc.math(5, nn[:, 1].copy(), None)
        u = (f32(1) - ((phi + PI).astype(f32) / (f32(2) * PI)).astype(f32)).astype(f32)
        v = ((theta + PI / f32(2)).astype(f32) / PI).astype(f32)
        ci = (orc.math(6, (u * f32(freq)).astype(f32), None) * f32(w - 1)).astype(f32)
        cj = ((f32(1) - orc.math(6, (v * f32(freq)).astype(f32), None)).astype(f32) * f32(h - 1)).astype(f32)

        def texel(c, mx):
            r = np.where(c >= f32(mx), mx, np.where(c > 0, np.floor(np.where(np.isfinite(c), c, 0)), 0))
            return np.where(c > 0, r, 0).astype(np.int64)  # texel_index: !(f > 0) -> 0, f >= max -> max, else truncate
        want = np.stack([texel(ci, w - 1), texel(cj, h - 1)], axis=1)
    bad = np.nonzero((exact != want).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} chain texels differ from the oracle's math, first normal {nn[bad[0]]!r}: {exact[bad[0]]} vs {want[bad[0]]}"
    share = fast[: n // 5].mean()  # on uniformly random directions
    assert not fast[-len(special):][[0, 1, 6, 7, 8, 9, 13]].any()  # the poles, the zero vector, NaNs, |y| > 1 and infinities go through the chain
    if freq * max(w, h) < 20_000:
        assert share > 0.97, share


def test_camera_quotients_through_the_reciprocal_are_exact(lib):
    """render.hpp:96-97: (x + xi) / width through div_exact with RN(1/width) — equal to the IEEE quotient for every frame
    size up to 16384 and numerators x + xi with xi a multiple of 2^-32 (checked on random and extreme numerators)."""
    rng = np.random.default_rng(3)
    for w in (1, 2, 3, 7, 225, 400, 480, 800, 1080, 1920, 2160, 3840, 5431, 16383, 16384):
        xs = rng.integers(0, w, 200_000).astype(np.float32)
        xi = (rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float32)
        xi[:4] = [0.0, 2.0 ** -32, 1.0 - 2.0 ** -24, 1.0]
        xs[:4] = [0, 0, w - 1, w - 1]
        n = (xs + xi).astype(np.float32)
        d = np.full_like(n, np.float32(w))
        assert_bit_identical(gpu_math(lib, 9, n, d), (n / d).astype(np.float32), f"(x + xi) / {w}")


def test_camera_rays_bit_exact(lib, orc):
    rng = np.random.default_rng(11)
    for cam_args, (w, h) in [(S.cornell_scene()[1], (1920, 1080)), (S.mixed_scene()[1], (400, 225))]:
        cam = scenes.make_camera(cam_args, w, h)
        n = 20_000
        xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
        st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        st[:4] = [0, 1, 0xFFFFFFFF, 2463534242]
        out = (abi.PtCameraRay * n)()
        abi.check(lib.pt_debug_camera_rays(C.byref(cam.c), w, h, xy.ctypes.data_as(C.POINTER(C.c_int32)),
                                           st.ctypes.data_as(C.POINTER(C.c_uint32)), out, n), "pt_debug_camera_rays")
        ref = orc.camera_rays(cam.c, w, h, xy, st)
        assert bytes(out) == bytes(ref)


def test_pinhole_camera_shortcut_is_exact(lib, orc):
    """aperture 0 and no zero among the origin's components: the device skips the lens arithmetic (camera_ray, pt_device.hpp) —
    same rays, same generator state after.  Cameras that must NOT take the shortcut sit beside: a zero origin component
    (origin + (-0) is +0, not the origin, when the origin is -0), negative zeros, a tiny aperture; and the signs of the axes
    vary so that the skipped products would be zeros of every sign."""
    rng = np.random.default_rng(12)
    w, h = 333, 187
    cams = []
    for frm, at, ap in [((278, 278, -800), (278, 278, 0), 0.0), ((-3.5, 2.25, 7.0), (1.0, -2.0, -4.0), 0.0), ((5.0, -1.0, 2.0), (-1.0, 3.0, 9.0), 0.0),
                        ((0.0, 1.0, 5.0), (0.0, 1.0, 0.0), 0.0), ((-0.0, -0.0, 5.0), (0.0, 0.0, 0.0), 0.0), ((3.0, 0.0, -0.0), (0.0, 0.0, 1.0), 0.0),
                        ((278, 278, -800), (278, 278, 0), 1e-30), ((1.0, 2.0, 3.0), (0.0, 0.0, 0.0), 0.5)]:
        cams.append(dict(look_from=frm, look_at=at, vup=(0, 1, 0), vfov=40.0, aperture=ap, focus_dist=10.0, time0=0.0, time1=1.0))
    for cam_args in cams:
        cam = scenes.make_camera(cam_args, w, h)
        n = 20_000
        xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
        st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        out = (abi.PtCameraRay * n)()
        abi.check(lib.pt_debug_camera_rays(C.byref(cam.c), w, h, xy.ctypes.data_as(C.POINTER(C.c_int32)),
                                           st.ctypes.data_as(C.POINTER(C.c_uint32)), out, n), "pt_debug_camera_rays")
        assert bytes(out) == bytes(orc.camera_rays(cam.c, w, h, xy, st)), cam_args
    # and through the render kernels (lane_regenerate): a pinhole camera whose origin has a zero keeps the general path
    ps, _ = S.cornell_scene()
    for frm in ((278, 278, -800), (0.0, 278, -800)):
        c = scenes.make_camera(dict(cams[0], look_from=frm), 64, 36)
        orc.set_math(True)
        assert_bit_identical(R.render_host(64, 36, 8, ps, c), orc.render(ps, c.c, 64, 36, 8), f"pinhole from {frm}")


def random_bounce_inputs(rng, n, center, extent):
    recs = (abi.PtBounceIn * n)()
    o = (rng.random((n, 3), dtype=np.float32) - 0.5) * 2 * extent + center
    target = (rng.random((n, 3), dtype=np.float32) - 0.5) * extent + center
    d = (target - o).astype(np.float32)
    d[::7] *= np.float32(0.01)   # short directions: t far beyond 1
    d[::11] *= np.float32(50.0)
    tm = rng.random(n, dtype=np.float32)
    st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    att = rng.random((n, 3), dtype=np.float32)
    for k in range(n):
        recs[k].origin[:] = o[k].tolist()
        recs[k].dir[:] = d[k].tolist()
        recs[k].time = float(tm[k])
        recs[k].rng_state = int(st[k])
        recs[k].attenuation[:] = att[k].tolist()
    return recs


BOUNCE_FIELDS = [n for n, _ in abi.PtBounceOut._fields_]


def compare_bounce(got, ref, n, what, has_image):
    for name in BOUNCE_FIELDS:
        if name in ("u", "v") and not has_image:
            continue  # u,v are only read by image textures; the kernel tracks them only then (pt_render.h)
        g = np.array([np.ctypeslib.as_array(getattr(got[k], name)) if hasattr(getattr(got[k], name), "__len__") else getattr(got[k], name) for k in range(n)])
        r = np.array([np.ctypeslib.as_array(getattr(ref[k], name)) if hasattr(getattr(ref[k], name), "__len__") else getattr(ref[k], name) for k in range(n)])
        if g.dtype.kind == "f":
            assert_bit_identical(g.astype(np.float32), r.astype(np.float32), f"{what}.{name}")
        else:
            bad = np.argwhere(g != r)
            assert len(bad) == 0, f"{what}.{name}: {len(bad)} differ, first {bad[0]}: {g[tuple(bad[0])]} vs {r[tuple(bad[0])]}"


@pytest.mark.parametrize("name,center,extent", [("cornell", (278, 278, 278), 700.0), ("mixed", (0, 0.3, -1), 3.0),
                                                ("spheres", (0, 0.5, 0), 5.0), ("triangles", (0, 1.5, 0), 4.0),
                                                ("ties", (0, 0, -2), 3.0), ("empty", (0, 0, 0), 1.0),
                                                ("badouel", (0, 0.3, -2), 3.0), ("sphere_ties", (0, 0.2, -2), 3.0),
                                                ("absorbed_ties", (0, 0.2, -2), 3.5), ("sphere_field", (0, 0.4, 0), 9.0)])
def test_bounce_bit_exact(lib, orc, name, center, extent):
    """hit_world + emitted + scatter of one ray (render.hpp:58-89): every hit_record field, the scattered
    ray, the attenuation and the RNG state after, for thousands of random rays."""
    ps, _ = S.ALL[name]()
    ds = R.DeviceScene(ps)
    rng = np.random.default_rng(hash(name) % 2 ** 31)
    n = 6000
    recs = random_bounce_inputs(rng, n, np.float32(center), np.float32(extent))
    out = (abi.PtBounceOut * n)()
    abi.check(lib.pt_debug_bounce(ds.handle, recs, out, n), "pt_debug_bounce")
    orc.set_math(True)
    ref = orc.bounce(ps, recs)
    statuses = [ref[k].status for k in range(n)]
    assert len(set(statuses)) >= (1 if name == "empty" else 2), "inputs should exercise hits and misses"
    has_image = any(ps.textures[i].kind == abi.PT_TEX_IMAGE for i in range(ps.n_textures))
    compare_bounce(out, ref, n, name, has_image)


def test_ties_across_absorbed_sphere_runs(lib, orc):
    """pt_flatten.hpp "absorbed sphere runs": a long sphere run's lists also test the static spheres of later short sphere runs, i.e. before
    the rect / box / triangle runs between them.  Rays aimed at the points where those surfaces tie EXACTLY (a sphere's pole in a plane; the
    same sphere twice) — straight down the z axis from several distances and with several direction lengths, plus a cloud around each —
    must resolve as the reference's list-order scan does: the oracle's hittable, t and scattered ray for every one, with the merge and
    without it (PtTuning.sphere_merge = -1), through the LDS kernels, the scalar-cache kernels and the bounce probe."""
    ps, cam = S.ALL["absorbed_ties"]()
    rays = []
    for (px, py) in ((0.0, 0.0), (1.2, 0.0), (-1.2, 0.0), (-2.4, 0.0), (0.1, 0.1), (1.2, 0.2), (-1.2, 0.1)):
        for oz in (1.0, 0.5, 2.0, 3.0):
            for s_ in (1.0, 2.0, 0.5, 4.0):
                rays.append(((px, py, oz), (0.0, 0.0, -s_)))
    rng = np.random.default_rng(66)
    for (px, py) in ((0.0, 0.0), (1.2, 0.0), (-1.2, 0.0), (-2.4, 0.0)):
        for _ in range(400):
            o = np.array([px, py, 1.0]) + rng.normal(size=3) * 0.05
            tgt = np.array([px, py, -1.5]) + rng.normal(size=3) * np.array([0.2, 0.2, 0.0])
            rays.append((tuple(o), tuple(tgt - o)))
    n = len(rays)
    recs = (abi.PtBounceIn * n)()
    for k, (o, d) in enumerate(rays):
        recs[k].origin[:] = [float(x) for x in o]; recs[k].dir[:] = [float(x) for x in d]; recs[k].time = 0.0
        recs[k].rng_state = 12345 + k; recs[k].attenuation[:] = [1.0, 1.0, 1.0]
    orc.set_math(True)
    ref = orc.bounce(ps, recs)
    tied = sum(1 for k in range(112) if ref[k].status != abi.PT_BOUNCE_MISS)
    assert tied >= 100, "the axis rays should hit the tied surfaces"
    # what the reference's scan returns at the engineered ties: the rect (hittable 4) over sphere A, the box (5) over C', the triangle (10) over B', F (3) over F'
    assert {ref[k].hittable for k in range(0, 16)} == {4} and {ref[k].hittable for k in range(16, 32)} == {5}
    # (the triangle's t comes out of other arithmetic than a plane's: at some distances it is an ulp off the sphere's and B' wins outright)
    assert {ref[k].hittable for k in range(32, 48)} == {10, 11} and {ref[k].hittable for k in range(48, 64)} == {3}
    for merge in (0, -1):
        ds = R.DeviceScene(ps, abi.tuning(sphere_merge=merge))
        out = (abi.PtBounceOut * n)()
        abi.check(lib.pt_debug_bounce(ds.handle, recs, out, n), "pt_debug_bounce")
        compare_bounce(out, ref, n, f"absorbed ties, sphere_merge {merge}", False)
        c = scenes.make_camera(cam, 96, 54)
        want = orc.render(ps, c.c, 96, 54, 16)
        for flags in (0, abi.PT_FLAG_NO_LDS, abi.PT_FLAG_FORCE_COOP, abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_PIXEL_GRANULAR):
            assert_bit_identical(R.render_host(96, 54, 16, ds, c, flags=flags), want, f"absorbed ties frame, sphere_merge {merge}, flags {flags}")


@pytest.mark.parametrize("name", list(S.ALL))
def test_golden_framebuffers(name):
    g = np.load(GOLDEN / f"fb_{name}_32x18x4.npy")
    ps, cam = S.ALL[name]()
    fb = R.render_host(32, 18, 4, ps, scenes.make_camera(cam, 32, 18))
    assert_bit_identical(fb, g, name)


@pytest.mark.parametrize("name,w,h,spp", [("cornell", 160, 90, 32), ("mixed", 128, 72, 24), ("spheres", 128, 72, 16),
                                          ("triangles", 96, 54, 8), ("ties", 64, 64, 16), ("empty", 50, 30, 4),
                                          ("mixed", 37, 21, 5)])
def test_framebuffer_bit_exact(orc, name, w, h, spp):
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    fb = R.render_host(w, h, spp, ps, c)
    orc.set_math(True)
    assert_bit_identical(fb, orc.render(ps, c.c, w, h, spp), f"{name} {w}x{h}x{spp}")


@pytest.mark.parametrize("name,w,h,spp", [("ties", 64, 64, 16), ("mixed", 128, 72, 24), ("cornell", 96, 54, 16), ("badouel", 64, 36, 8)])
def test_small_scenes_with_slab_pools_forced(orc, lib, name, w, h, spp):
    """The small hand-made scenes — coincident rect / box faces (`ties`), media and triangles between rects (`mixed`) — with a
    slab pool for every stretch of two or more rects / boxes (PT_POOL_ALWAYS; the cost model builds none for scenes this
    small): framebuffer and thousands of random rays, bit-identical."""
    import os
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    os.environ["PT_POOL_ALWAYS"] = "1"
    try:
        ds = R.DeviceScene(ps)
    finally:
        del os.environ["PT_POOL_ALWAYS"]
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    for what, flags in (("LDS", abi.PT_FLAG_NO_COOP), ("scalar cache", abi.PT_FLAG_NO_LDS), ("default", 0)):
        assert_bit_identical(R.render_host(w, h, spp, ds, c, flags=flags), ref, f"{name} pools forced, {what}")
    rng = np.random.default_rng(5)
    centre, extent = {"ties": ((0, 0, -2), 3.0), "mixed": ((0, 0.3, -1), 3.0), "cornell": ((278, 278, 278), 700.0), "badouel": ((0, 0.3, -2), 3.0)}[name]
    recs = random_bounce_inputs(rng, 6000, np.float32(centre), np.float32(extent))
    out = (abi.PtBounceOut * 6000)()
    abi.check(lib.pt_debug_bounce(ds.handle, recs, out, 6000), "pt_debug_bounce")
    compare_bounce(out, orc.bounce(ps, recs), 6000, name + " pools forced", False)


def test_depth_edge_cases(orc):
    ps, cam = S.spheres_scene()
    c = scenes.make_camera(cam, 40, 24)
    orc.set_math(True)
    for depth in (0, 1, 2, 7):
        assert_bit_identical(R.render_host(40, 24, 3, ps, c, depth=depth), orc.render(ps, c.c, 40, 24, 3, depth=depth), f"depth {depth}")


def test_config1_smoke_scene_400x225x64(orc):
    """BASELINE.json configs[0] shape: the main.cpp scene (496 hittables), 400x225, 64 spp, depth 50."""
    ps, cam = scenes.build("smoke")
    c = scenes.make_camera(cam, 400, 225)
    fb = R.render_host(400, 225, 64, ps, c)
    orc.set_math(True)
    ref = orc.render(ps, c.c, 400, 225, 64)
    assert_bit_identical(fb, ref, "smoke 400x225x64")
    # against the reference's libm semantics on this host: statistical (tolerance stated in the module docstring)
    orc.set_math(False)
    ref_libm = orc.render(ps, c.c, 400, 225, 64)
    p = psnr_8bit(orc.tonemap_rgb8(fb), orc.tonemap_rgb8(ref_libm))
    assert p >= 38.0, f"PSNR vs glibc oracle {p:.1f} dB"  # BASELINE.md §4's bar; measured 83.5 dB (99.8 % of the pixels bit-identical)
    assert abs(fb.mean() / ref_libm.mean() - 1) < 0.02


@pytest.mark.parametrize("name", ["mixed", "spheres"])
def test_vs_glibc_oracle_statistical(orc, name):
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, 96, 54)
    fb = R.render_host(96, 54, 16, ps, c)
    orc.set_math(False)
    ref = orc.render(ps, c.c, 96, 54, 16)
    p = psnr_8bit(orc.tonemap_rgb8(fb), orc.tonemap_rgb8(ref))
    assert p >= 38.0, f"{name}: PSNR {p:.1f} dB"  # BASELINE.md §4's bar; measured: identical 8-bit images
    assert abs(np.nanmean(fb) / np.nanmean(ref) - 1) < 0.02


def test_cornell_bit_exact_vs_glibc_oracle(orc):
    """No transcendental on the Cornell path: the GPU frame equals the libm-mode oracle bit for bit too."""
    ps, cam = S.cornell_scene()
    c = scenes.make_camera(cam, 200, 112)
    orc.set_math(False)
    assert_bit_identical(R.render_host(200, 112, 64, ps, c), orc.render(ps, c.c, 200, 112, 64))


@pytest.mark.parametrize("name", ["cornell", "mixed", "triangles", "sphere_ties"])
def test_lds_and_scalar_fetch_agree(name):
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, 64, 40)
    a = R.render_host(64, 40, 8, ps, c)
    b = R.render_host(64, 40, 8, ps, c, flags=abi.PT_FLAG_NO_LDS)
    assert_bit_identical(a, b, name)


@pytest.mark.parametrize("name", ["cornell", "mixed", "triangles", "spheres", "ties", "sphere_ties", "badouel", "absorbed_ties"])
def test_streaming_kernel_agrees(name):
    """The LDS-tile streaming kernel (used when the scene exceeds LDS) gives the resident kernel's frame."""
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, 64, 40)
    a = R.render_host(64, 40, 8, ps, c)
    b = R.render_host(64, 40, 8, ps, c, flags=abi.PT_FLAG_FORCE_STREAM)
    assert_bit_identical(a, b, name)


@pytest.mark.parametrize("name", ["cornell", "spheres", "triangles", "ties", "mixed", "sphere_ties", "absorbed_ties"])
@pytest.mark.parametrize("size", [(1, 1), (2, 1), (3, 1), (5, 1), (3, 3), (17, 1), (8, 4), (33, 1), (40, 2)])
def test_streaming_kernel_cooperative_tail(orc, name, size):
    """The streaming kernel spreads the rays of a wave that is down to <= 32 live lanes over groups of G = 64 >>
    ceil(log2 live) lanes (G = 64, 32, 16, 8, 4, 2 for these frame sizes; 33 and 80 pixels: ordinary scan, then the tail)
    — bit-identical to the oracle, ties included ("mixed" has media: it must take the ordinary scan)."""
    w, h = size
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, 6)
    assert_bit_identical(R.render_host(w, h, 6, ps, c, flags=abi.PT_FLAG_FORCE_STREAM | abi.PT_FLAG_PIXEL_GRANULAR), ref, f"{name} {w}x{h}")
    assert_bit_identical(R.render_host(w, h, 6, ps, c, flags=abi.PT_FLAG_FORCE_STREAM | abi.PT_FLAG_NO_COOP), ref, f"{name} {w}x{h} no coop")


def test_streaming_kernel_cooperative_tail_many_tiles(orc):
    """3 000 triangles (several LDS tiles) with few live rays: the strided scan across tile boundaries + small runs either
    side, against the oracle."""
    ps, cam = S.triangles_scene(3000)
    orc.set_math(True)
    for w, h in ((3, 2), (16, 2), (48, 27)):
        c = scenes.make_camera(cam, w, h)
        assert_bit_identical(R.render_host(w, h, 5, ps, c, flags=abi.PT_FLAG_FORCE_STREAM), orc.render(ps, c.c, w, h, 5), f"{w}x{h}")


def test_sphere_runs_masks_chunks_and_mixed_shutter_intervals(orc):
    """Sphere runs longer than a 32-sphere mask word, with moving spheres at every position class (first, last, chunk
    borders), in every kernel family — and the same scene with a second shutter interval, which takes the
    one-sphere-at-a-time fallback of sphere_scan."""
    from path_tracer_amd.scene import lambertian_material, metal_material, pack, sphere
    rng = scenes.HostRNG(99)
    def build(second_interval):
        hs = [sphere((0, -1000, 0), 1000, lambertian_material((0.5, 0.5, 0.5)))]
        for i in range(70):
            c = (-3.5 + 0.1 * i + 0.05 * float(rng.float_t()), 0.2 + 0.3 * float(rng.float_t()), -2.0 + 4.0 * float(rng.float_t()))
            mat = lambertian_material(tuple(rng.vec_t())) if i % 3 else metal_material(tuple(rng.vec_t(0.5, 1)), 0.2)
            if i in (0, 5, 30, 31, 32, 33, 63, 64, 69) or i % 7 == 3:
                t0, t1 = ((0.25, 0.75) if (second_interval and i == 33) else (0.0, 1.0))
                hs.append(sphere(c, (c[0], c[1] + 0.4, c[2]), t0, t1, 0.15, mat))
            else:
                hs.append(sphere(c, 0.15, mat))
        return pack(hs)
    cam = dict(look_from=(0, 1.5, 6), look_at=(0, 0.3, 0), vup=(0, 1, 0), vfov=45.0, aperture=0.05, focus_dist=6.0, time0=0.0, time1=1.0)
    orc.set_math(True)
    for second in (False, True):
        ps = build(second)
        c = scenes.make_camera(cam, 72, 40)
        ref = orc.render(ps, c.c, 72, 40, 6)
        for flags in (0, abi.PT_FLAG_FORCE_COOP, abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_NO_LDS, abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT):
            assert_bit_identical(R.render_host(72, 40, 6, ps, c, flags=flags), ref, f"second interval {second}, flags {flags}")


def test_streaming_kernel_sphere_tiles(orc):
    """A sphere run longer than one LDS tile (672 spheres) through the streaming kernel: the mask word index follows the
    tile's position in the run."""
    from path_tracer_amd.scene import lambertian_material, pack, sphere
    rng = scenes.HostRNG(5)
    hs = []
    for i in range(1500):
        c = (-4 + 8 * float(rng.float_t()), 0.1 + 2 * float(rng.float_t()), -4 + 8 * float(rng.float_t()))
        m = lambertian_material(tuple(rng.vec_t()))
        hs.append(sphere(c, (c[0], c[1] + 0.2, c[2]), 0.0, 1.0, 0.05, m) if i % 5 == 2 else sphere(c, 0.05, m))
    ps = pack(hs)
    cam = dict(look_from=(0, 2, 9), look_at=(0, 1, 0), vup=(0, 1, 0), vfov=50.0, aperture=0.0, focus_dist=9.0, time0=0.0, time1=1.0)
    c = scenes.make_camera(cam, 40, 24)
    orc.set_math(True)
    ref = orc.render(ps, c.c, 40, 24, 4)
    assert_bit_identical(R.render_host(40, 24, 4, ps, c, flags=abi.PT_FLAG_FORCE_STREAM), ref, "streamed sphere tiles")
    assert_bit_identical(R.render_host(40, 24, 4, ps, c), ref, "resident")


def test_streaming_kernel_many_tiles(orc):
    """3 000 triangles = 9 000 records: several LDS tiles per run, partial last tile, small runs either side."""
    ps, cam = S.triangles_scene(3000)
    c = scenes.make_camera(cam, 48, 27)
    fb = R.render_host(48, 27, 3, ps, c, flags=abi.PT_FLAG_FORCE_STREAM)
    orc.set_math(True)
    assert_bit_identical(fb, orc.render(ps, c.c, 48, 27, 3), "streamed 3000 triangles")
    assert_bit_identical(fb, R.render_host(48, 27, 3, ps, c, flags=abi.PT_FLAG_NO_LDS), "stream vs scalar")


@pytest.mark.parametrize("name", ["cornell", "mixed", "triangles"])
def test_dequeue_granularity_does_not_matter(name):
    """Which lane renders which pixel (tile-granular or pixel-granular dequeue, persistent grid) cannot change a
    pixel: its RNG stream is seeded by its own global id."""
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, 100, 60)
    a = R.render_host(100, 60, 12, ps, c, flags=abi.PT_FLAG_TILE_GRANULAR)
    b = R.render_host(100, 60, 12, ps, c, flags=abi.PT_FLAG_PIXEL_GRANULAR)
    assert_bit_identical(a, b, name)
    s_ = R.render_host(100, 60, 12, ps, c, flags=abi.PT_FLAG_PIXEL_GRANULAR | abi.PT_FLAG_FORCE_STREAM)
    assert_bit_identical(a, s_, name + " streamed")


@pytest.mark.parametrize("name,flags", [("mixed", 0), ("cornell", 0), ("mixed", abi.PT_FLAG_PIXEL_GRANULAR),
                                        ("triangles", abi.PT_FLAG_FORCE_STREAM), ("cornell", abi.PT_FLAG_FORCE_COOP),
                                        ("spheres", abi.PT_FLAG_FORCE_COOP),
                                        ("spheres", abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_PIXEL_GRANULAR)])
def test_cost_sorted_tile_order_does_not_matter(orc, name, flags):
    """spp >= 16 and >= 64 tiles turn on the cost-probe pass + heaviest-first tile order; it decides when a pixel
    is rendered, never its value."""
    ps, cam = S.ALL[name]()
    w, h, spp = 120, 72, 32  # 15 x 9 = 135 tiles
    c = scenes.make_camera(cam, w, h)
    a = R.render_host(w, h, spp, ps, c, flags=flags)
    b = R.render_host(w, h, spp, ps, c, flags=flags | abi.PT_FLAG_NO_LPT)
    assert_bit_identical(a, b, name)
    orc.set_math(True)
    assert_bit_identical(a, orc.render(ps, c.c, w, h, spp), name + " vs oracle")


@pytest.mark.parametrize("name", ["cornell", "spheres", "triangles", "ties", "mixed", "sphere_ties", "absorbed_ties"])
@pytest.mark.parametrize("size", [(37, 21), (64, 40), (9, 5)])
def test_cooperative_traversal_agrees(orc, name, size):
    """Waves with <= 32 live lanes split each ray's list over idle lanes and merge the segment winners with the
    list-order tie rule; odd frame sizes make partially filled tiles, so this path runs from the first iteration.
    ('mixed' puts an image texture on a triangle, which disables the path: it must then be a no-op.)"""
    w, h = size
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    a = R.render_host(w, h, 24, ps, c, flags=abi.PT_FLAG_FORCE_COOP)
    b = R.render_host(w, h, 24, ps, c, flags=abi.PT_FLAG_NO_COOP)
    assert_bit_identical(a, b, f"{name} {w}x{h}")
    assert_bit_identical(a, R.render_host(w, h, 24, ps, c, flags=abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT), "no split")
    orc.set_math(True)
    assert_bit_identical(a, orc.render(ps, c.c, w, h, 24), f"{name} {w}x{h} vs oracle")


@pytest.mark.parametrize("name", ["cornell", "spheres", "triangles", "ties", "mixed", "sphere_ties", "absorbed_ties"])
def test_wide_phase_every_group_size(orc, name, monkeypatch):
    """Heavy tiles are rendered G lanes per pixel: all G lanes hold the same pixel (same seed, same draws), each tests the
    hittables == its lane (mod G), a butterfly merges the partial winners with the scan's own tie rule.  Tuning knobs force
    EVERY tile through that phase for every group size; the frame must not change by a bit.  ('ties' has coincident faces
    of different kinds: the tie rule is what is being tested there.)"""
    w, h = 136, 72  # 17 x 9 = 153 tiles (>= 64 and spp >= 16: the probe pass that feeds the split runs)
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    ref = R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_NO_COOP)
    orc.set_math(True)
    assert_bit_identical(ref, orc.render(ps, c.c, w, h, 16), f"{name} ordinary vs oracle")
    monkeypatch.setenv("PT_SPLIT_TILES", "-1")
    for log_g in range(1, 7):
        monkeypatch.setenv("PT_WIDE_LOGG", str(log_g))
        assert_bit_identical(R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_FORCE_COOP), ref, f"{name} G={1 << log_g}")
    monkeypatch.setenv("PT_SPLIT_TILES", "40")  # mixed launch: 40 tiles wide, the rest ordinary, waves change phase
    monkeypatch.setenv("PT_WIDE_LOGG", "2")
    assert_bit_identical(R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_FORCE_COOP), ref, f"{name} 40 tiles wide")


def test_schedule_probe_reports_the_wide_phase(lib, monkeypatch):
    """pt_debug_schedule: what the makespan model decided for the last render (tiles through the wide phase, lanes per pixel)."""
    ps, cam = scenes.build("smoke")
    ds = R.DeviceScene(ps)
    c = scenes.make_camera(cam, 136, 72)
    out = (C.c_int32 * 2)()
    R.render(136, 72, 16, ds, c, flags=abi.PT_FLAG_NO_COOP)
    assert lib.pt_debug_schedule(ds.handle, out) == 0 and (out[0], out[1]) == (0, 0)  # ordinary kernel: no such phase
    monkeypatch.setenv("PT_SPLIT_TILES", "40")
    monkeypatch.setenv("PT_WIDE_LOGG", "4")
    ds2 = R.DeviceScene(ps)  # the tuning knobs are read when a scene is created, never on the launch path
    R.render(136, 72, 16, ds2, c, flags=abi.PT_FLAG_FORCE_COOP)  # (by default a scene with a sphere grid runs the grid kernels)
    assert lib.pt_debug_schedule(ds2.handle, out) == 0 and (out[0], out[1]) == (40, 16)
    R.render(136, 72, 16, ds, c, flags=abi.PT_FLAG_FORCE_COOP)  # the scene created before the knobs were set still follows the model
    assert lib.pt_debug_schedule(ds.handle, out) == 0 and 0 <= out[0] <= 153 and out[1] in (0, 2, 4, 8, 16, 32, 64)
    monkeypatch.delenv("PT_SPLIT_TILES"); monkeypatch.delenv("PT_WIDE_LOGG")


def test_wide_phase_smoke_scene(orc, monkeypatch):
    """496 hittables with a constant_medium suffix and image textures: the medium is scanned after the merge by every lane
    of a group (identical RNG state in all of them); u,v travel through the butterfly."""
    ps, cam = scenes.build("smoke")
    w, h = 96, 56
    c = scenes.make_camera(cam, w, h)
    ref = R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_NO_COOP)
    monkeypatch.setenv("PT_SPLIT_TILES", "-1")
    for log_g in (1, 3, 5, 6):
        monkeypatch.setenv("PT_WIDE_LOGG", str(log_g))
        assert_bit_identical(R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_FORCE_COOP), ref, f"smoke G={1 << log_g}")


def test_cooperative_traversal_with_medium_suffix_and_image(orc):
    """The SmokeSphere scene: image textures (u,v carried through the merge), a constant_medium at the end of the
    list (scanned after the merge with the owner's RNG state), 496 hittables split over up to 64 lanes."""
    ps, cam = scenes.build("smoke")
    for w, h in ((19, 11), (40, 24)):
        c = scenes.make_camera(cam, w, h)
        a = R.render_host(w, h, 16, ps, c)  # 496 hittables: the launcher picks the cooperative kernel by itself
        assert_bit_identical(a, R.render_host(w, h, 16, ps, c, flags=abi.PT_FLAG_NO_COOP), f"smoke {w}x{h}")
        orc.set_math(True)
        assert_bit_identical(a, orc.render(ps, c.c, w, h, 16), f"smoke {w}x{h} vs oracle")


@pytest.mark.parametrize("name", ["cornell", "ties", "spheres"])
def test_lds_resident_cold_lane_state_agrees(orc, name, monkeypatch):
    """Small scenes without image textures keep each lane's radiance sum / sample count / pixel id in LDS slots; the
    register-resident variant of the same kernel must give the same bits (and both the oracle's)."""
    ps, cam = S.ALL[name]()
    for (w, h, spp) in ((96, 54, 20), (13, 9, 3)):
        c = scenes.make_camera(cam, w, h)
        a = R.render_host(w, h, spp, ps, c)
        monkeypatch.setenv("PT_NO_COLD_LDS", "1")
        b = R.render_host(w, h, spp, ps, c)
        monkeypatch.delenv("PT_NO_COLD_LDS")
        assert_bit_identical(a, b, f"{name} {w}x{h}")
        orc.set_math(True)
        assert_bit_identical(a, orc.render(ps, c.c, w, h, spp), f"{name} {w}x{h} vs oracle")


def test_image_texture_uv_modes(orc):
    """u,v reach image textures three ways: derived from the final hit (scene whose image textures sit on spheres, rects or
    boxes only), tracked through the scan ('mixed': an image texture on a triangle inherits stale values), or not at all."""
    from path_tracer_amd.scene import (TextureAtlas, sphere, xy_rect, xz_rect, yz_rect, box, lambertian_material,
                                       lightsource_material, image_texture, pack)
    rng = np.random.default_rng(5)
    atlas = TextureAtlas()
    tex = image_texture.from_array(rng.integers(0, 256, size=(16, 32, 3), dtype=np.uint8), 3.0, atlas)
    m_img = lambertian_material(tex)
    grey = lambertian_material((0.6, 0.6, 0.6))
    hs = [sphere((0, -100.5, -1), 100, grey), sphere((0, 0, -1), 0.5, m_img), xy_rect(-2, -0.8, -0.5, 1, -1.5, m_img),
          box((0.8, -0.5, -1.6), (1.6, 0.4, -0.9), m_img), yz_rect(-0.5, 1, -2, 0, -2.2, lightsource_material(tex)),
          xz_rect(-3, 3, -3, 1, 2.5, lightsource_material((2, 2, 2)))]
    ps = pack(hs, atlas)
    cam = dict(look_from=(0.3, 0.6, 1.5), look_at=(0, 0, -1), vup=(0, 1, 0), vfov=60.0, aperture=0.0, focus_dist=2.0,
               time0=0.0, time1=1.0)
    for (w, h, spp) in ((80, 48, 12), (136, 72, 16)):
        c = scenes.make_camera(cam, w, h)
        a = R.render_host(w, h, spp, ps, c)
        orc.set_math(True)
        assert_bit_identical(a, orc.render(ps, c.c, w, h, spp), f"image textures on sphere/rect/box {w}x{h}")
        assert_bit_identical(a, R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_FORCE_COOP), "cooperative kernel")
        assert_bit_identical(a, R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_FORCE_STREAM), "streaming kernel")
        assert_bit_identical(a, R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_NO_LDS), "scalar-cache kernel")


@pytest.mark.parametrize("name", ["cornell", "mixed", "ties"])
def test_plain_division_path_agrees(name):
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, 64, 40)
    assert_bit_identical(R.render_host(64, 40, 8, ps, c), R.render_host(64, 40, 8, ps, c, flags=abi.PT_FLAG_NO_FASTDIV), name)


def test_irregular_rays_take_the_plain_division(lib, orc):
    """Axis-parallel / zero / huge directions are not 'regular': they must still match the oracle bit for bit."""
    ps, _ = S.cornell_scene()
    ds = R.DeviceScene(ps)
    dirs = [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 0, -1), (1e-30, 0.5, 1), (1e30, 1, 1), (0, 0, 0), (np.nan, 0, 1),
            (np.inf, 1, 1), (1, 1, 1e-45), (0.0, -0.0, 1.0), (1e-41, 1e-41, 1e-41)]
    origins = [(278, 278, -800), (278, 278, 278), (0, 0, 0), (555, 555, 555), (278, 0, 278), (1e20, 0, 0)]
    recs = (abi.PtBounceIn * (len(dirs) * len(origins)))()
    k = 0
    for o in origins:
        for d in dirs:
            recs[k].origin[:] = [float(np.float32(v)) for v in o]
            recs[k].dir[:] = [float(np.float32(v)) for v in d]
            recs[k].time = 0.5
            recs[k].rng_state = 12345 + k
            recs[k].attenuation[:] = [1.0, 1.0, 1.0]
            k += 1
    out = (abi.PtBounceOut * k)()
    abi.check(lib.pt_debug_bounce(ds.handle, recs, out, k), "pt_debug_bounce")
    orc.set_math(True)
    compare_bounce(out, orc.bounce(ps, recs), k, "irregular", False)


@pytest.mark.parametrize("shards", [2, 3, 8])
@pytest.mark.parametrize("size", [(64, 40), (21, 13)])
def test_sharded_render_equals_full(torch_gpu, shards, size):
    """Tile sharding (one shard per GPU) + the root-side un-interleave reproduce the single-GPU frame."""
    torch = torch_gpu
    w, h = size
    ps, cam = S.mixed_scene()
    c = scenes.make_camera(cam, w, h)
    ds = R.DeviceScene(ps)
    full = R.render(w, h, 6, ds, c)
    parts = [R.render(w, h, 6, ds, c, shard_index=i, shard_count=shards) for i in range(shards)]
    gathered = torch.stack(parts)
    fb = R.unshard(gathered, w, h, shards)
    torch.cuda.synchronize()
    assert_bit_identical(fb.cpu().numpy(), full.cpu().numpy(), f"{shards} shards")
    assert_bit_identical(unshard_reference(gathered.cpu().numpy(), w, h, shards), full.cpu().numpy())


def test_tonemap_matches_output_stage(torch_gpu, orc):
    torch = torch_gpu
    ps, cam = S.mixed_scene()
    c = scenes.make_camera(cam, 48, 27)
    fb = R.render(48, 27, 4, ps, c)
    fb[0, 0, 0] = float("nan"); fb[0, 1, 1] = -1.0; fb[1, 0, 2] = float("inf")
    rgb = R.tonemap_rgb8(fb)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rgb.cpu().numpy(), orc.tonemap_rgb8(fb.cpu().numpy()))


def test_full_size_cornell_1080p_sampled_pixels(torch_gpu, orc):
    """BASELINE.json configs[1] at FULL size (1920x1080, 1024 spp, depth 50): the whole frame on the GPU,
    then 1500 sampled pixels re-rendered by the oracle at full spp and compared bit for bit (a pixel's value
    depends only on its own RNG stream, so sampled pixels check the full-size run exactly)."""
    torch = torch_gpu
    w, h, spp = 1920, 1080, 1024
    ps, cam = scenes.build("cornell")
    c = scenes.make_camera(cam, w, h)
    fb, ms = R.render(w, h, spp, ps, c, timed=True)
    fbn = fb.cpu().numpy()
    assert np.isfinite(fbn).all()
    rng = np.random.default_rng(2024)
    xy = np.stack([rng.integers(0, w, 1500), rng.integers(0, h, 1500)], axis=1).astype(np.int32)
    xy[:6] = [[0, 0], [w - 1, h - 1], [0, h - 1], [w - 1, 0], [960, 540], [959, 1079]]
    orc.set_math(True)
    ref = orc.render_pixels(ps, c.c, w, h, spp, xy)
    assert_bit_identical(fbn[xy[:, 1], xy[:, 0]], ref, "1080p x 1024spp sampled pixels")
    print(f"\n[cornell 1080p 1024spp] kernel {ms:.1f} ms = {w * h * spp / ms / 1e3:.1f} Msamples/s")


def test_full_size_smoke_1080p_sampled_pixels(torch_gpu, orc):
    """BASELINE.json configs[2] at FULL size (the 496-hittable scene with the reference's two image textures,
    1920x1080, 1024 spp, depth 50): cooperative kernel, cost-sorted order and split queue all active; 400 sampled pixels
    re-rendered by the oracle at full spp, bit for bit."""
    w, h, spp = 1920, 1080, 1024
    ps, cam = scenes.build("smoke")
    c = scenes.make_camera(cam, w, h)
    fb, ms = R.render(w, h, spp, ps, c, timed=True)
    fbn = fb.cpu().numpy()
    rng = np.random.default_rng(7)
    xy = np.stack([rng.integers(0, w, 400), rng.integers(0, h, 400)], axis=1).astype(np.int32)
    orc.set_math(True)
    assert_bit_identical(fbn[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, w, h, spp, xy), "smoke 1080p sampled pixels")
    print(f"\n[cfg3 smoke 1080p {spp}spp] kernel {ms:.1f} ms = {w * h * spp / ms / 1e3:.1f} Msamples/s")


@pytest.mark.parametrize("shard", [0, 5])
def test_cfg4_shard_of_8_at_4k_4096spp(torch_gpu, orc, shard):
    """BASELINE.json configs[3] (SmokeSphere, 3840x2160, 4096 spp, 8-way tile-sharded): what ONE of its eight GPUs
    renders — shard `shard` of 8, full resolution, full spp — with sampled pixels of that shard re-rendered by the
    oracle at full spp, bit for bit; the shard's tile layout is checked through the same sampled pixels."""
    w, h, spp, n = 3840, 2160, 4096, 8
    ps, cam = scenes.build("smoke")
    c = scenes.make_camera(cam, w, h)
    local, ms = R.render(w, h, spp, ps, c, shard_index=shard, shard_count=n, timed=True)
    tiles = local.cpu().numpy()  # [tiles_per_shard][64][3]
    tiles_x = (w + 7) // 8
    rng = np.random.default_rng(40 + shard)
    lt = rng.integers(0, tiles.shape[0] - 1, 160)  # local tile l is global tile l * n + shard (pt_render.h)
    inner = rng.integers(0, 64, 160)
    g = lt * n + shard
    xy = np.stack([(g % tiles_x) * 8 + (inner & 7), (g // tiles_x) * 8 + (inner >> 3)], axis=1).astype(np.int32)
    assert (xy[:, 0] < w).all() and (xy[:, 1] < h).all()
    orc.set_math(True)
    assert_bit_identical(tiles[lt, inner], orc.render_pixels(ps, c.c, w, h, spp, xy), f"cfg4 shard {shard}/8 sampled pixels")
    samples = tiles.shape[0] * 64 * spp
    print(f"\n[cfg4 shard {shard}/8 of 4K x {spp}spp] kernel {ms:.1f} ms = {samples / ms / 1e3:.1f} Msamples/s on this GPU")


def test_full_size_triangle_mesh_1080p_sampled_pixels(torch_gpu, orc):
    """BASELINE.json configs[4] at FULL size: 100 000 triangles (through the triangle pool: fine grid + direction maps, round 5) + ground
    sphere + emissive rect, 1920x1080, 256 spp.  256 sampled pixels are re-rendered by the oracle at full spp (was 64)."""
    ps, cam = scenes.build("triangles", n_triangles=100_000)
    w, h, spp = 1920, 1080, 256
    c = scenes.make_camera(cam, w, h)
    fb, ms = R.render(w, h, spp, R.DeviceScene(ps), c, timed=True)
    fbn = fb.cpu().numpy()
    rng = np.random.default_rng(99)
    xy = np.stack([rng.integers(0, w, 256), rng.integers(0, h, 256)], axis=1).astype(np.int32)
    orc.set_math(True)
    assert_bit_identical(fbn[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, w, h, spp, xy), "100k triangles 1080p sampled pixels")
    print(f"\n[cfg5 triangles 1080p {spp}spp] kernel {ms:.1f} ms = {w * h * spp / ms / 1e3:.2f} Msamples/s")


def test_a_mesh_of_1_5_million_triangles_keeps_its_pool(torch_gpu, orc, lib):
    """rounds 3-4 kept the pool's tables in the blob (24-bit record offsets: no pool beyond ~830 k triangles, no scene
    beyond 5.5 M).  A 1.5 M-triangle mesh is pooled (tables in their own buffer, 25-bit offsets in hit ids; its direction maps within the
    default budget at whatever resolution fits) and a low-spp frame is bit-exact on sampled pixels against the oracle's full scan."""
    import ctypes as C
    ps, cam = scenes.triangle_mesh_scene(n_triangles=1_500_000)
    st = (C.c_int32 * 8)()
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert st[0] == 1_500_000 and st[6] > 4_500_000            # pooled; the blob alone is 4.5 M records (> 2^22, fine below 2^25)
    w, h, spp = 480, 270, 2
    c = scenes.make_camera(cam, w, h)
    fb, ms = R.render(w, h, spp, R.DeviceScene(ps), c, timed=True)
    fbn = fb.cpu().numpy()
    rng = np.random.default_rng(5)
    xy = np.stack([rng.integers(0, w, 48), rng.integers(0, h, 48)], axis=1).astype(np.int32)
    orc.set_math(True)
    assert_bit_identical(fbn[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, w, h, spp, xy), "1.5 M triangles sampled pixels")
    print(f"\n[1.5 M triangles {w}x{h}x{spp}] kernel {ms:.1f} ms; pool maps (K entries) {list(st)[2:4]} at {(st[4] >> 20) & 1023} / {(st[4] >> 10) & 1023} / {st[4] & 1023}")


def test_cfg1_reference_textures_full_frame_and_png(torch_gpu, orc, tmp_path):
    """BASELINE.json configs[0]: the default SmokeSphere scene with the REAL image textures (decoded Xilinx.jpg /
    SYCL.png from tests/golden/cfg1_textures.npz at the atlas offsets of texture.hpp:113-114), 400x225, 64 spp: every
    pixel bit-exact against the oracle, and out.png written through the device output stage (pt_tonemap_rgb8) + png.py
    decodes to the oracle's 8-bit image (main.cpp:33-59)."""
    import zlib
    from path_tracer_amd.png import write_png
    torch = torch_gpu
    w, h, spp = 400, 225, 64
    ps, cam = scenes.build("smoke")
    assert len(ps.atlas) == 3 + 1024 * 512 * 3 + 1280 * 559 * 3
    c = scenes.make_camera(cam, w, h)
    fb = R.render(w, h, spp, ps, c)
    rgb8 = R.tonemap_rgb8(fb)
    torch.cuda.synchronize()
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    assert_bit_identical(fb.cpu().numpy(), ref, "cfg1 with the reference textures")
    np.testing.assert_array_equal(rgb8.cpu().numpy(), orc.tonemap_rgb8(ref))
    out = tmp_path / "out.png"
    write_png(out, rgb8.cpu().numpy())
    raw = out.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    idat, pos = b"", 8
    while pos < len(raw):  # minimal PNG reader: concatenate IDAT, undo filter type 0 rows
        ln = int.from_bytes(raw[pos:pos + 4], "big"); kind = raw[pos + 4:pos + 8]
        if kind == b"IDAT":
            idat += raw[pos + 8:pos + 8 + ln]
        pos += 12 + ln
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 3)
    assert (rows[:, 0] == 0).all()
    np.testing.assert_array_equal(rows[:, 1:].reshape(h, w, 3), orc.tonemap_rgb8(ref))
    # the image textures really are on screen: the logo sphere (top left) shows the SYCL logo's yellow and red
    top = orc.tonemap_rgb8(ref)[:60, 60:140].reshape(-1, 3).astype(int)
    assert ((top[:, 0] > 150) & (top[:, 1] > 150) & (top[:, 2] < 110)).sum() > 200


def test_full_size_1080p_full_frame_low_spp(orc):
    """Every pixel of a 1920x1080 frame (2 spp): seeds up to 2,073,599, edge tiles, all tile rows."""
    ps, cam = scenes.build("cornell")
    c = scenes.make_camera(cam, 1920, 1080)
    orc.set_math(True)
    assert_bit_identical(R.render_host(1920, 1080, 2, ps, c), orc.render(ps, c.c, 1920, 1080, 2), "1080p x 2spp")


def test_missing_image_falls_back_to_texel_zero(orc, capsys):
    """texture.hpp:106-111: a texture file that cannot be loaded prints an error and becomes a 1x1 image at atlas
    offset 0 = the {0,0,1} fallback texel."""
    from path_tracer_amd.scene import TextureAtlas, image_texture, lambertian_material, lightsource_material, pack, sphere, xy_rect
    atlas = TextureAtlas()
    t = image_texture.image_texture_factory("/nonexistent/texture.png", 3.0, atlas)
    assert (t.width, t.height, t.offset) == (1, 1, 0)
    assert "Could not load texture image file" in capsys.readouterr().err
    good = image_texture.from_array(np.full((4, 4, 3), 200, np.uint8), 1.0, atlas)  # forces the atlas to be shipped
    hs = [sphere((0, 0, -1), 0.5, lightsource_material(t)), xy_rect(-2, 2, -2, 2, -3, lambertian_material(good)),
          sphere((0, -100.5, -1), 100, lambertian_material(t))]
    ps = pack(hs, atlas)
    cam = dict(look_from=(0, 0, 1), look_at=(0, 0, -1), vup=(0, 1, 0), vfov=60.0, aperture=0.0, focus_dist=1.0, time0=0.0, time1=0.0)
    c = scenes.make_camera(cam, 40, 24)
    fb = R.render_host(40, 24, 8, ps, c)
    orc.set_math(True)
    assert_bit_identical(fb, orc.render(ps, c.c, 40, 24, 8), "fallback texel")
    centre = fb[12, 20]
    assert centre[0] == 0 and centre[1] == 0 and abs(centre[2] - 1 / 255) < 1e-6  # the light shows texel {0,0,1}/255


@pytest.mark.parametrize("w,h,spp", [(1, 1, 9), (1, 70, 5), (70, 1, 5), (7, 7, 33), (9, 9, 17), (513, 3, 4)])
def test_degenerate_frame_sizes(orc, w, h, spp):
    """Frames smaller than a tile, one pixel wide/high, just over a tile: padding lanes, partial tiles, and the
    pixel whose generator is stuck at 0 (render.hpp:131)."""
    ps, cam = S.spheres_scene()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp)
    for flags in (0, abi.PT_FLAG_FORCE_COOP, abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_NO_LDS):
        assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"{w}x{h} flags {flags}")


def test_4k_frame_low_spp(orc):
    """BASELINE.json configs[3] frame size (3840x2160): every pixel at 1 spp (seeds up to 8,294,399)."""
    ps, cam = scenes.build("cornell")
    c = scenes.make_camera(cam, 3840, 2160)
    orc.set_math(True)
    assert_bit_identical(R.render_host(3840, 2160, 1, ps, c), orc.render(ps, c.c, 3840, 2160, 1), "4K x 1spp")


@pytest.mark.parametrize("name,w,h,spp", [("cornell", 64, 40, 200), ("mixed", 48, 32, 64), ("spheres", 40, 24, 130),
                                          ("triangles", 33, 17, 65), ("sphere_ties", 32, 18, 20)])
def test_fast_mode_matches_its_own_oracle(orc, name, w, h, spp):
    """PT_FLAG_FAST_RNG — opt-in, NOT the reference's image: per-(pixel, chunk) RNG streams (pt_fast_seed), chunks of 64
    samples summed in chunk order.  Deterministic, so it has a bit-exact checker of its own: the oracle's restatement of
    the same mode.  Whole chunks, a short last chunk, fewer samples than one chunk; resident, streaming and sharded."""
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    F = abi.PT_FLAG_FAST_RNG
    ref = orc.render(ps, c.c, w, h, spp, flags=F)
    assert not np.array_equal(ref, orc.render(ps, c.c, w, h, spp))  # it really is a different image
    for flags in (F, F | abi.PT_FLAG_FORCE_STREAM, F | abi.PT_FLAG_NO_LDS, F | abi.PT_FLAG_NO_LPT):
        assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"fast mode {name} flags {flags}")
    shard = R.render_host(w, h, spp, ps, c, flags=F, shard_index=1, shard_count=3)
    assert_bit_identical(shard, orc.render(ps, c.c, w, h, spp, shard_index=1, shard_count=3, flags=F), "fast mode shard 1/3")


@pytest.mark.parametrize("name,w,h,spp,ref_spp", [("cornell", 96, 54, 128, 4096), ("smoke", 96, 54, 64, 2048)])
def test_fast_mode_is_statistically_the_parity_image(orc, name, w, h, spp, ref_spp):
    """The tolerance of the opt-in fast mode, stated: against a converged image (the oracle at 32x the samples) the fast
    frame and the parity frame are estimates of equal quality — their 8-bit PSNRs differ by < 0.75 dB (measured 0.1-0.4)
    and the mean radiance of each is within 1.5 % of the converged mean."""
    ps, cam = scenes.build(name)
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, ref_spp)
    parity = R.render_host(w, h, spp, ps, c)
    fast = R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_FAST_RNG)
    r8 = orc.tonemap_rgb8(ref)
    pp, pf = psnr_8bit(orc.tonemap_rgb8(parity), r8), psnr_8bit(orc.tonemap_rgb8(fast), r8)
    assert abs(pp - pf) < 0.75, f"PSNR vs converged: parity {pp:.2f} dB, fast {pf:.2f} dB"
    assert abs(fast.mean() / ref.mean() - 1) < 0.015 and abs(parity.mean() / ref.mean() - 1) < 0.015
    print(f"\n[fast mode {name}] PSNR vs converged: parity {pp:.2f} dB, fast {pf:.2f} dB")


@pytest.mark.parametrize("name,w,h,spp", [("mixed", 24, 14, 3), ("cornell", 20, 12, 4), ("spheres", 18, 10, 3), ("sphere_ties", 16, 9, 2),
                                          ("triangles", 12, 7, 2), ("empty", 5, 3, 2)])
def test_single_stream_executor(orc, lib, name, w, h, spp):
    """PT_FLAG_SINGLE_STREAM = the reference's USE_SINGLE_TASK executor (render.hpp:113-122): ONE default-seeded RNG stream
    for the whole frame, pixels x-outer / y-inner — a different image from the parallel executor's, bit-exact against the
    oracle's restatement of it.  Sequential by definition: one lane, small frames only, no shards."""
    ps, cam = S.ALL[name]()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    ref = orc.render(ps, c.c, w, h, spp, flags=abi.PT_FLAG_SINGLE_STREAM)
    if name != "empty":
        assert not np.array_equal(ref, orc.render(ps, c.c, w, h, spp))
    assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=abi.PT_FLAG_SINGLE_STREAM), ref, f"single stream {name}")
    ds = R.DeviceScene(ps)
    fb = np.zeros(4096 * 4096 * 3, np.float32)
    big = abi.PtRenderParams(4096, 4096, 1, 50, 0, 1, abi.PT_FLAG_SINGLE_STREAM, 0)
    assert lib.pt_render_host(ds.handle, C.byref(c.c), C.byref(big), fb.ctypes.data_as(FP)) == abi.PT_ERR_TOO_LARGE
    sharded = abi.PtRenderParams(w, h, spp, 50, 0, 2, abi.PT_FLAG_SINGLE_STREAM, 0)
    assert lib.pt_render_host(ds.handle, C.byref(c.c), C.byref(sharded), fb.ctypes.data_as(FP)) == abi.PT_ERR_INVALID_ARG


def test_badouel_strategy_triangles(orc, lib):
    """_triangle<badouel_ray_triangle_intersec> (triangle.hpp:14-56; PtHittable.strategy = PT_TRI_BADOUEL): its own device
    kind and kernel instantiations; bit-exact against the oracle through the resident, streaming and single-stream paths,
    sharded, and with the cooperative / fast paths politely out of the way."""
    ps, cam = S.badouel_scene()
    orc.set_math(True)
    for (w, h, spp) in ((96, 54, 12), (9, 5, 6)):
        c = scenes.make_camera(cam, w, h)
        ref = orc.render(ps, c.c, w, h, spp)
        for flags in (0, abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_FORCE_COOP, abi.PT_FLAG_NO_LPT, abi.PT_FLAG_PIXEL_GRANULAR):
            assert_bit_identical(R.render_host(w, h, spp, ps, c, flags=flags), ref, f"badouel {w}x{h} flags {flags}")
    c = scenes.make_camera(cam, 40, 24)
    assert_bit_identical(R.render_host(40, 24, 5, ps, c, shard_index=2, shard_count=3),
                         orc.render(ps, c.c, 40, 24, 5, shard_index=2, shard_count=3), "badouel shard 2/3")
    assert_bit_identical(R.render_host(14, 8, 2, ps, c := scenes.make_camera(cam, 14, 8), flags=abi.PT_FLAG_SINGLE_STREAM),
                         orc.render(ps, c.c, 14, 8, 2, flags=abi.PT_FLAG_SINGLE_STREAM), "badouel single stream")
    # the two strategies are different arithmetic for the same geometry: close images, not identical ones
    mt = S.badouel_scene()[0]
    for i in range(mt.n_hittables):
        mt.hittables[i].strategy = 0
    c = scenes.make_camera(cam, 96, 54)
    a, b = orc.render(ps, c.c, 96, 54, 12), orc.render(mt, c.c, 96, 54, 12)
    assert not np.array_equal(a, b) and psnr_8bit(orc.tonemap_rgb8(a), orc.tonemap_rgb8(b)) > 20.0
    ds = R.DeviceScene(ps)
    p = abi.PtRenderParams(16, 8, 4, 50, 0, 1, abi.PT_FLAG_FAST_RNG, 0)
    fb = np.zeros(16 * 8 * 3, np.float32)
    assert lib.pt_render_host(ds.handle, C.byref(c.c), C.byref(p), fb.ctypes.data_as(FP)) == abi.PT_ERR_INVALID_ARG


def test_heaviest_tiles_in_narrow_pieces_change_nothing(orc, torch_gpu):
    """PtTuning.heavy_tiles (pt_render.hip: launch / lane_acquire): the head of the cost-sorted tile queue handed out 16 pixels at a time
    — a quarter tile per wave, for launches bound by their heaviest tiles' chains — is another partition of the same pixels: the frame
    must equal the whole-tile frame bit for bit and the oracle's on sampled pixels, for head lengths below, at and beyond the tile count
    (and for a shard, where the launcher's own rule applies the mode)."""
    import torch
    ps, cam = S.sphere_field_scene()
    orc.set_math(True)
    W, H, spp = 1280, 600, 16  # 12 000 tiles, 1.7 pixels per resident lane (this small scene runs seven workgroups per CU): grid kernels, whole tiles, cost probe — the rule's own range
    c = scenes.make_camera(cam, W, H)
    frames = {}
    for t in (-1, 0, 7, 64, 100000):
        ds = R.DeviceScene(ps, tuning=abi.tuning(heavy_tiles=t))
        frames[t] = R.render(W, H, spp, ds, c).cpu().numpy()
        torch.cuda.synchronize()
        ll = (C.c_int32 * 4)()
        abi.check(abi.load_library().pt_debug_last_launch(ds.handle, ll), "pt_debug_last_launch")
        want = {-1: 0, 0: 1024 * 64, 7: 7 * 64, 64: 64 * 64, 100000: (W // 8) * (H // 8) // 2 * 64}[t]
        assert ll[1] == 64 and ll[2] == want, (t, list(ll))  # whole tiles + the narrow head of the expected length
    for t in (0, 7, 64, 100000):
        assert_bit_identical(frames[t], frames[-1], f"heavy_tiles={t} against whole tiles")
    xy = np.stack([np.random.default_rng(5).integers(0, W, 1500), np.random.default_rng(6).integers(0, H, 1500)], axis=1).astype(np.int32)
    assert_bit_identical(frames[64][xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, W, H, spp, xy), "narrow head against the oracle")
    # a shard in the rule's own range (pixels per resident lane between 1.5 and 6 needs a big frame: forced here, rule-checked in tools/heavy_probe.py)
    a = R.render(W, H, spp, R.DeviceScene(ps, tuning=abi.tuning(heavy_tiles=32)), c, shard_index=1, shard_count=3)
    b = R.render(W, H, spp, R.DeviceScene(ps, tuning=abi.tuning(heavy_tiles=-1)), c, shard_index=1, shard_count=3)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_the_cost_probes_samples_are_the_frames_first_samples(orc, torch_gpu):
    """Round 5 (pt_render.hip: KArgs.resume_rng): the probe pass that orders the tiles leaves every pixel's radiance sum in the framebuffer
    and its generator state in a workspace, and the frame launch carries on from both at sample `probe_spp` — the reference's single
    stream per pixel and its order of additions (render.hpp:95-101), no sample rendered twice.  The frame must equal the one that throws
    the probe away (PtTuning.probe_resume = -1), the one without a probe (PT_FLAG_NO_LPT) and the oracle's, bit for bit: whole frames,
    shards (compact tile buffers), the three kernel families (slab pools with LDS-resident cold state, sphere grid, cooperative lists)."""
    import torch
    orc.set_math(True)
    cases = [("cornell", scenes.build("cornell"), 256, 192, 64),   # headline family: cold lane state in LDS
             ("field", S.sphere_field_scene(), 640, 400, 32),        # grid kernels: the probe keeps the heaviest pixel, 4 000 tiles
             ("mixed", (S.mixed_scene()[0], S.mixed_scene()[1]), 192, 128, 48)]
    for name, (ps, cam), W, H, spp in cases:
        c = scenes.make_camera(cam, W, H)
        kept = R.render(W, H, spp, R.DeviceScene(ps), c)
        again = R.render(W, H, spp, R.DeviceScene(ps, tuning=abi.tuning(probe_resume=-1)), c)
        plain = R.render(W, H, spp, R.DeviceScene(ps), c, flags=abi.PT_FLAG_NO_LPT)
        assert torch.equal(kept.view(torch.int32), again.view(torch.int32)), name
        assert torch.equal(kept.view(torch.int32), plain.view(torch.int32)), name
        xy = np.stack([np.random.default_rng(11).integers(0, W, 300), np.random.default_rng(12).integers(0, H, 300)], axis=1).astype(np.int32)
        assert_bit_identical(kept.cpu().numpy()[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, W, H, spp, xy), f"{name}: resumed frame against the oracle")
        a = R.render(W, H, spp, R.DeviceScene(ps), c, shard_index=1, shard_count=3)
        b = R.render(W, H, spp, R.DeviceScene(ps, tuning=abi.tuning(probe_resume=-1)), c, shard_index=1, shard_count=3)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name


def test_longest_remaining_chain_first_changes_no_value(orc, torch_gpu):
    """Round 5 (pt_render.hip: render_kernel, KArgs.prio_*): from the middle of the tile queue on, the headline family's waves set their issue
    priority by the samples their slowest pixel still has to render.  Only WHEN a pixel is rendered changes: the frame must equal the one
    rendered without priorities (PtTuning.chain_priority = -1) and the oracle's, bit for bit — whole frames and shards, long enough for the
    rule to act (more than 64 iterations per wave after half the queue is taken)."""
    import torch
    orc.set_math(True)
    ps, cam = scenes.build("cornell")
    for W, H, spp, shard in ((512, 384, 128, None), (640, 360, 64, (1, 3))):
        c = scenes.make_camera(cam, W, H)
        kw = dict(shard_index=shard[0], shard_count=shard[1]) if shard else {}
        on = R.render(W, H, spp, R.DeviceScene(ps), c, **kw)
        off = R.render(W, H, spp, R.DeviceScene(ps, tuning=abi.tuning(chain_priority=-1)), c, **kw)
        assert torch.equal(on.view(torch.int32), off.view(torch.int32)), (W, H, spp, shard)
        if not shard:
            xy = np.stack([np.random.default_rng(21).integers(0, W, 200), np.random.default_rng(22).integers(0, H, 200)], axis=1).astype(np.int32)
            assert_bit_identical(on.cpu().numpy()[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, W, H, spp, xy), "priorities on, against the oracle")


def test_launcher_rules_for_chain_bound_launches(torch_gpu):
    """What launch_render decides, read back through pt_debug_last_launch: the headline family keeps four workgroups per CU when a launch
    has fewer than 1.15 tiles per wave slot (a shard of 4 - 6 of the 1080p frame; 1.6 and shards of 3 on before the chain priorities of round 5)
    and its full occupancy otherwise; the 496-hittable scene's
    whole 1080p frame runs whole tiles through the queued walk, its shard 0/3 whole tiles with the narrow head, its shard 0/8 the
    lanes_cap regime.  (The images under these rules are covered by the parity tests; this pins the decisions.)"""
    import torch
    lib = abi.load_library()
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def decided(ds, cam, n):
        R.render(1920, 1080, 16, ds, cam, shard_index=0, shard_count=n)
        torch.cuda.synchronize()
        ll = (C.c_int32 * 4)()
        abi.check(lib.pt_debug_last_launch(ds.handle, ll), "pt_debug_last_launch")
        return list(ll)

    packed, cam_args = scenes.build("cornell")
    cam, ds = scenes.make_camera(cam_args, 1920, 1080), None
    ds = R.DeviceScene(packed)
    full = decided(ds, cam, 1)[0]
    assert full >= 6 * cus and decided(ds, cam, 2)[0] == full          # whole frame and halves: every wave slot
    assert decided(ds, cam, 4)[0] == 4 * cus and decided(ds, cam, 3)[0] == full
    assert decided(ds, cam, 8)[0] <= 4 * cus                             # half a tile per slot: one wave per tile
    packed, cam_args = scenes.build("smoke")
    cam, ds = scenes.make_camera(cam_args, 1920, 1080), R.DeviceScene(packed)
    assert decided(ds, cam, 1)[1:] == [64, 0, 1]
    assert decided(ds, cam, 3)[1:] == [64, 4 * cus * 64, 1]
    one_of_8 = decided(ds, cam, 8)
    assert one_of_8[1] == 16 and one_of_8[2] == 0


@pytest.mark.parametrize("walk", [1, 2])
def test_sphere_grid_is_exact(orc, monkeypatch, walk):
    """(walk: PtTuning.grid_walk — 1 the wave-synchronous walk, 2 the walk through the LDS pair queue; the launcher would pick by frame.)
    The culling grid of sphere runs (pt_flatten.hpp: build_sphere_grid; pt_device.hpp: sphere_grid_walk) against the
    oracle's linear scan, bit for bit: the ordinary view; a camera 3 000 units away (primary rays start beyond the grid's
    rlimit: the wave takes the full lists, bounces come back to the grid); a shutter wider than the spheres' interval (ray
    times outside [time0, time1]: full lists); from inside the field looking along it (long walks); every kernel family."""
    monkeypatch.setenv("PT_GRID_WALK", str(walk))
    ps, cam = S.sphere_field_scene()
    orc.set_math(True)
    views = [dict(cam), dict(cam, look_from=(2000.0, 900.0, 2000.0), vfov=0.4, focus_dist=3000.0, aperture=0.0),
             dict(cam, time0=-0.5, time1=1.5), dict(cam, look_from=(-5.5, 0.4, -5.5), look_at=(6, 0.3, 6), vfov=70.0, aperture=0.0)]
    for vi, view in enumerate(views):
        c = scenes.make_camera(view, 64, 36)
        ref = orc.render(ps, c.c, 64, 36, 6)
        # NO_COOP / NO_LDS: the kernels that walk the grid (a frame this small would otherwise keep the cooperative ones)
        for flags in (abi.PT_FLAG_NO_COOP, abi.PT_FLAG_NO_LDS, abi.PT_FLAG_NO_COOP | abi.PT_FLAG_PIXEL_GRANULAR, 0, abi.PT_FLAG_FORCE_COOP,
                      abi.PT_FLAG_FORCE_STREAM, abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT):
            assert_bit_identical(R.render_host(64, 36, 6, ps, c, flags=flags), ref, f"view {vi} flags {flags}")
    c = scenes.make_camera(cam, 48, 27)
    fast = R.render_host(48, 27, 70, ps, c, flags=abi.PT_FLAG_FAST_RNG)
    assert_bit_identical(fast, orc.render(ps, c.c, 48, 27, 70, flags=abi.PT_FLAG_FAST_RNG), "grid + fast mode")
    # a frame large enough for the launcher to pick the grid kernels by itself (>= 4096 tiles), sampled pixels
    c = scenes.make_camera(cam, 640, 520)
    big = R.render_host(640, 520, 4, ps, c)
    xy = np.stack([np.random.default_rng(1).integers(0, 640, 3000), np.random.default_rng(2).integers(0, 520, 3000)], axis=1).astype(np.int32)
    assert_bit_identical(big[xy[:, 1], xy[:, 0]], orc.render_pixels(ps, c.c, 640, 520, 4, xy), "640x520 through the launcher's own choice")
    monkeypatch.setenv("PT_NO_GRID", "1")  # read at scene creation: the brute-force lists for the same scene
    c = scenes.make_camera(cam, 64, 36)
    assert_bit_identical(R.render_host(64, 36, 6, ps, c, flags=abi.PT_FLAG_NO_COOP), orc.render(ps, c.c, 64, 36, 6), "PT_NO_GRID")


def test_scene_reserve_then_render(orc, lib):
    """pt_scene_reserve sizes the launch workspaces ahead of time (the LPT arrays, the fast mode's partial sums); renders after
    it give the same frames; bad arguments are codes."""
    ps, cam = S.spheres_scene()
    c = scenes.make_camera(cam, 136, 80)
    ds = R.DeviceScene(ps)
    ds.reserve(136, 80, 64, flags=abi.PT_FLAG_FAST_RNG)
    ds.reserve(136, 80, 64)
    orc.set_math(True)
    assert_bit_identical(R.render_host(136, 80, 64, ds, c), orc.render(ps, c.c, 136, 80, 64), "after reserve")
    F = abi.PT_FLAG_FAST_RNG
    assert_bit_identical(R.render_host(136, 80, 64, ds, c, flags=F), orc.render(ps, c.c, 136, 80, 64, flags=F), "fast mode after reserve")
    bad = abi.PtRenderParams(0, 8, 1, 50, 0, 1, 0, 0)
    assert lib.pt_scene_reserve(ds.handle, C.byref(bad)) == abi.PT_ERR_INVALID_ARG
    assert lib.pt_scene_reserve(None, C.byref(abi.PtRenderParams(8, 8, 1, 50, 0, 1, 0, 0))) == abi.PT_ERR_INVALID_ARG


def test_rerender_is_deterministic(torch_gpu):
    ps, cam = scenes.build("smoke")
    c = scenes.make_camera(cam, 200, 112)
    ds = R.DeviceScene(ps)
    a = R.render(200, 112, 8, ds, c).cpu().numpy()
    b = R.render(200, 112, 8, ds, c).cpu().numpy()
    assert_bit_identical(a, b)


def test_errors_are_codes_not_crashes(lib):
    ps, cam = S.cornell_scene()
    c = scenes.make_camera(cam, 8, 8)
    ds = R.DeviceScene(ps)
    p = abi.PtRenderParams(8, 8, 0, 50, 0, 1, 0, 0)
    fb = np.zeros(8 * 8 * 3, np.float32)
    assert lib.pt_render_host(ds.handle, C.byref(c.c), C.byref(p), fb.ctypes.data_as(FP)) == abi.PT_ERR_INVALID_ARG
    assert lib.pt_render_host(None, C.byref(c.c), C.byref(p), fb.ctypes.data_as(FP)) == abi.PT_ERR_INVALID_ARG
    ps.hittables[0].material = 77
    with pytest.raises(abi.PtError) as e:
        R.DeviceScene(ps)
    assert e.value.code == abi.PT_ERR_BAD_SCENE


def test_more_launches_in_flight_than_queue_ring_slots(torch_gpu, orc):
    """pt_render is asynchronous; each launch takes a slot of the scene's ring of dequeue counters (kQueueRing = 256).  600
    back-to-back asynchronous 8x8 renders into 600 framebuffers on one scene and one stream (two slots each: none — the
    frame is too small for a probe — so 600 slots, the ring wraps twice) must all be the oracle's frame: a wrapped ring
    waits for the launch that last used the slot instead of sharing its counter."""
    torch = torch_gpu
    ps, cam = S.cornell_scene()
    c = scenes.make_camera(cam, 8, 8)
    ds = R.DeviceScene(ps)
    orc.set_math(True)
    ref = orc.render(ps, c.c, 8, 8, 24)
    outs = torch.empty((600, 8, 8, 3), dtype=torch.float32, device="cuda")
    for i in range(600):
        R.render(8, 8, 24, ds, c, out=outs[i])
    torch.cuda.synchronize()
    got = outs.cpu().numpy()
    for i in (0, 1, 255, 256, 257, 511, 512, 599):
        assert_bit_identical(got[i], ref, f"launch {i}")
    assert (got.view(np.uint32) == got[0].view(np.uint32)[None]).all()


def test_two_streams_get_two_device_scenes(torch_gpu, orc):
    """render() keeps one DeviceScene per (PackedScene, device, STREAM): a PtScene's launch workspaces (tile costs / order)
    are per scene, so concurrent renders from two streams must not share one."""
    torch = torch_gpu
    ps, cam = scenes.build("smoke", textures="procedural")
    c = scenes.make_camera(cam, 320, 184)
    orc.set_math(True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    frames = []
    for rep in range(3):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                frames.append(R.render(320, 184, 64, ps, c))
    torch.cuda.synchronize()
    assert len(ps.__dict__["_pt_device_scenes"]) == 2
    xy = np.stack([np.random.default_rng(4).integers(0, 320, 300), np.random.default_rng(5).integers(0, 184, 300)], axis=1).astype(np.int32)
    ref = orc.render_pixels(ps, c.c, 320, 184, 64, xy)
    for k, f in enumerate(frames):
        assert_bit_identical(f.cpu().numpy()[xy[:, 1], xy[:, 0]], ref, f"frame {k}")


def test_device_scene_cache_is_bounded(torch_gpu, orc):
    """render()'s per-PackedScene cache of device scenes is keyed by (device, stream handle); programs that create
    streams on the fly must not keep a full device copy of the scene per stream that ever existed — it is a small LRU, and an
    evicted scene lives on exactly as long as a frame rendered from it does."""
    torch = torch_gpu
    ps, cam = S.ALL["mixed"]()
    c = scenes.make_camera(cam, 40, 24)
    orc.set_math(True)
    ref = orc.render(ps, c.c, 40, 24, 3)
    frames = []
    streams = [torch.cuda.Stream() for _ in range(7)]
    for st in streams:
        with torch.cuda.stream(st):
            frames.append(R.render(40, 24, 3, ps, c))
    torch.cuda.synchronize()
    assert len(ps.__dict__["_pt_device_scenes"]) <= 4
    for k, f in enumerate(frames):
        assert_bit_identical(f.cpu().numpy(), ref, f"stream {k}")
