"""The inequalities the triangle pool's exactness rests on (path_tracer_amd/csrc/pt_tripool.hpp), checked on a binary32 emulation
of the reference's Moller-Trumbore test (triangle.hpp:58-100: same operations, same order, no contraction) for random rays and
for adversarial rays that lie almost in a triangle's plane:

  every (ray, triangle) pair the emulated test accepts is either GRAZING by the pool's own band test
  (|d . N'_i| < |d| (rho P_i + Q_i): it is then a band / always-list candidate), or NOT grazing and then
  (a) the exact line-plane point, clamped onto the forward ray, lies inside the triangle's box grown by sigma'_i
      (so the grid cell that contains that point lists the triangle), and
  (b) the computed t is within kappa (relative) + 1.2 L_i / ((M - 1) |d|) of the exact parameter (so the walk's range
      [0, closest (1 + kappa)] reaches that cell);
  and in every case the ray's LINE passes within the radius the two distance filters allow of the centroid.

The GPU suite then checks the walked structure end to end against the oracle, bit for bit (tests/test_gpu_fuzz.py)."""
import ctypes as C

import numpy as np

from path_tracer_amd import abi, scenes
from path_tracer_amd.scene import hittable_dtype

f32 = np.float32
U = 2.0 ** -24
M, MA, SAFE = 12.0, 256.0, 1.5   # pt_tripool.hpp: TriPoolTuning defaults and SAFE


def cross32(a, b):
    return np.stack([(a[..., 1] * b[..., 2]).astype(f32) - (a[..., 2] * b[..., 1]).astype(f32),
                     (a[..., 2] * b[..., 0]).astype(f32) - (a[..., 0] * b[..., 2]).astype(f32),
                     (a[..., 0] * b[..., 1]).astype(f32) - (a[..., 1] * b[..., 0]).astype(f32)], -1).astype(f32)


def dot32(a, b):
    return (((a[..., 0] * b[..., 0]).astype(f32) + (a[..., 1] * b[..., 1]).astype(f32)).astype(f32) + (a[..., 2] * b[..., 2]).astype(f32)).astype(f32)


def test_accepted_pairs_are_band_or_grid_candidates():
    ps, _ = scenes.triangle_mesh_scene(n_triangles=20_000)
    h = np.frombuffer(ps.hittables, dtype=hittable_dtype)
    f = h["f"][1:-1].astype(f32)
    v0 = f[:, 0:3]
    e1, e2 = (f[:, 3:6] - v0).astype(f32), (f[:, 6:9] - v0).astype(f32)
    e1d, e2d, v0d = e1.astype(np.float64), e2.astype(np.float64), v0.astype(np.float64)
    Nd = np.cross(e1d, e2d)
    l1, l2 = np.linalg.norm(e1d, axis=1), np.linalg.norm(e2d, axis=1)
    L = np.maximum(l1, l2)
    P = M * 17.5 * U * L * SAFE
    Q = (MA * 7 + 4) * U * l1 * l2 * SAFE + 2.0 ** -40
    sig = 8.5 * L / (M - 1)
    centre = v0d.mean(0)
    R = np.linalg.norm(v0d - centre, axis=1).max()
    lo = np.minimum(np.minimum(v0d, v0d + e1d), v0d + e2d)
    hi = np.maximum(np.maximum(v0d, v0d + e1d), v0d + e2d)
    cen = v0d + (e1d + e2d) / 3
    Np = Nd.astype(f32).astype(np.float64)
    kr = 6 * SAFE * U * L * L * (17.5 + 7 * L / R)
    rng = np.random.default_rng(5)
    stats = dict(accepted=0, band=0, grid=0)

    def check(o, d):
        o, d = o.astype(f32), d.astype(f32)
        hh = cross32(np.broadcast_to(d, e2.shape), e2)
        a = dot32(e1, hh)
        s = (o - v0).astype(f32)
        uu = dot32(s, hh)
        q = cross32(s, e1)
        v = dot32(np.broadcast_to(d, q.shape), q)
        w = dot32(e2, q)
        with np.errstate(all="ignore"):
            t = (w / a).astype(f32)
        aa = np.abs(a)
        ok = ~(aa < f32(1e-7)) & ~((uu > 0) != (a > 0)) & ~(np.abs(uu) > aa) & ~((v > 0) != (a > 0)) & ~(np.abs((uu + v).astype(f32)) > aa) & ~(t < f32(0.001))
        od, dd = o.astype(np.float64), d.astype(np.float64)
        dn = np.linalg.norm(dd)
        rho = np.linalg.norm(od - centre) + R
        ap = np.abs(Np @ dd)
        band = ap < dn * (rho * P + Q)
        for i in np.nonzero(ok)[0]:
            stats["accepted"] += 1
            # distance of the ray's line from the centroid
            dist_line = np.linalg.norm(np.cross(cen[i] - od, dd)) / dn
            if band[i]:
                stats["band"] += 1
                a1 = ap[i] - 4 * U * L[i] * L[i] * dn
                if a1 > 0:
                    assert dist_line <= L[i] + kr[i] * rho * dn / a1, ("noise-radius filter", i)
                continue
            stats["grid"] += 1
            a_ = -(dd @ Nd[i])
            th = (e2d[i] @ np.cross(od - v0d[i], e1d[i])) / a_
            Pp = od + max(th, 0.0) * dd
            assert np.max(np.maximum(np.maximum(lo[i] - Pp, Pp - hi[i]), 0)) <= sig[i], ("grown box", i)
            assert abs(float(t[i]) - th) <= 2.2 / (MA - 1) * abs(th) + 1.2 * L[i] / ((M - 1) * dn) + 1e-12, ("t", i)
            assert dist_line <= L[i] * (1 + 8.5 / (M - 1)), ("grid ball filter", i)

    for _ in range(60):
        o = rng.uniform([-3, 0, -3], [3, 3, 3])
        d = rng.normal(size=3)
        check(o, d / np.linalg.norm(d) * rng.uniform(0.2, 2))
    n = len(v0)
    for _ in range(400):   # rays that lie almost in the plane of a triangle and pass near it
        i = int(rng.integers(n))
        nh = Nd[i] / max(np.linalg.norm(Nd[i]), 1e-300)
        t1 = e1d[i] / l1[i]
        t2 = np.cross(nh, t1)
        ang = rng.uniform(0, 2 * np.pi)
        d = (np.cos(ang) * t1 + np.sin(ang) * t2 + nh * 10 ** rng.uniform(-8, -2) * rng.choice([-1, 1])) * rng.uniform(0.3, 2)
        target = v0d[i] + rng.uniform(-0.2, 1.2) * e1d[i] + rng.uniform(-0.2, 1.2) * e2d[i] + rng.normal(size=3) * 10 ** rng.uniform(-7, -3)
        check(target - d / np.linalg.norm(d) * rng.uniform(0.01, 12), d)
    assert stats["accepted"] > 3000 and stats["band"] > 20 and stats["grid"] > 2000, stats


def test_pool_thresholds_and_tables(lib, monkeypatch):
    """Long triangle runs (>= 4096) get a pool by default — the 100 k-triangle mesh of BASELINE config 5 does — shorter ones only
    with PT_TRICULL=1 (>= 256: the fuzz fields), none with PT_NO_TRICULL; every triangle sits in exactly one of {three band
    levels, always list}, and the blob grows by the inline candidate records."""
    ps, _ = scenes.triangle_mesh_scene()
    st = (C.c_int32 * 8)()
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert st[0] == 100_000 and st[1] + st[2] + st[3] + st[4] == 100_000
    assert 2000 < st[5] < 30000 and 1.5e7 < st[6] * 16 < 2.0e8
    n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
    abi.check(lib.pt_debug_flatten(C.byref(ps.desc), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
    assert flags.value & 4 and n_runs.value == 3
    small, _ = scenes.triangle_mesh_scene(n_triangles=1000)
    abi.check(lib.pt_debug_tri_pool(C.byref(small.desc), st), "pt_debug_tri_pool")
    assert st[0] == 0                       # 1000 triangles: full scan by default
    monkeypatch.setenv("PT_TRICULL", "1")
    abi.check(lib.pt_debug_tri_pool(C.byref(small.desc), st), "pt_debug_tri_pool")
    assert st[0] == 1000
    monkeypatch.setenv("PT_NO_TRICULL", "1")
    for sc in (ps, small):
        abi.check(lib.pt_debug_tri_pool(C.byref(sc.desc), st), "pt_debug_tri_pool")
        assert list(st)[:6] == [0] * 6
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert st[6] * 16 < 5.0e6               # the plain blob: 4.8 MB
