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
MG = 96.0                        # the grid's tight slack (TriPoolTuning::Mg)


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
    Pg = MG * 17.5 * U * L * SAFE                       # the grid's TIGHT radius: pairs with |a^| >= thr(MG)
    sig_g = (6 / MG + 6 / MA + 1.2 / (MG - 1)) * L
    centre = v0d.mean(0)
    R = np.linalg.norm(v0d - centre, axis=1).max()
    lo = np.minimum(np.minimum(v0d, v0d + e1d), v0d + e2d)
    hi = np.maximum(np.maximum(v0d, v0d + e1d), v0d + e2d)
    cen = v0d + (e1d + e2d) / 3
    rv = np.max(np.stack([np.linalg.norm(v0d - cen, axis=1), np.linalg.norm(v0d + e1d - cen, axis=1), np.linalg.norm(v0d + e2d - cen, axis=1)]), axis=0)
    Np = Nd.astype(f32).astype(np.float64)
    kr = 6 * SAFE * U * L * L * (17.5 + 7 * L / R)
    rng = np.random.default_rng(5)
    stats = dict(accepted=0, band=0, grid=0, tight=0, loose=0)

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
            # the two-radius filter of the compressed grid records: not grazing at MG -> the walked point P' within Rv + sigma'(MG)
            # of the centroid (tight); otherwise the pair passes the band test at MG by definition, and P' is within Rv + sigma'(M)
            # <= (Rv + sigma'(MG)) (1 + 2 (8.5 / (M - 1) - sigma'(MG) / L)) (loose: every edge is <= 2 Rv)
            assert L[i] <= 2 * rv[i] * (1 + 1e-12)
            if ap[i] >= dn * (rho * Pg[i] + Q[i]):
                stats["tight"] += 1
                assert np.max(np.maximum(np.maximum(lo[i] - Pp, Pp - hi[i]), 0)) <= sig_g[i], ("tight box", i)
                assert np.linalg.norm(Pp - cen[i]) <= rv[i] + sig_g[i], ("tight ball", i)
            else:
                stats["loose"] += 1
                assert np.linalg.norm(Pp - cen[i]) <= (rv[i] + sig_g[i]) * (1 + 2 * (8.5 / (M - 1) - sig_g[i] / L[i])), ("loose ball", i)

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
    assert stats["accepted"] > 3000 and stats["band"] > 20 and stats["grid"] > 2000 and stats["tight"] > 1500 and stats["loose"] > 50, stats


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


def test_meshes_too_large_for_a_pool_fall_back_to_the_plain_blob(lib):
    """the pool's tables ride in the blob (~17 records of 16 bytes per triangle beside the triangle's own 3) and
    record offsets are 24 bits, so a mesh of a million triangles — which flattens to 50 MB without a pool and rendered fine before
    round 3 — must not fail with PT_ERR_TOO_LARGE because the pool is on by default: it is flattened WITHOUT a pool."""
    ps, _ = scenes.triangle_mesh_scene(n_triangles=1_050_000)
    st = (C.c_int32 * 8)()
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert st[0] == 0                                           # no pool
    assert 1_050_000 * 3 < st[6] < 1_050_000 * 3 + 64           # the plain blob: three records per triangle + headers
    assert st[6] < (1 << 24)


def test_compressed_filter_records_round_to_the_safe_side(lib, monkeypatch):
    """The device streams QUANTISED filter records (pt_tripool.hpp "compressed records").  Parse them back out of the flattened
    blob of a 3000-triangle field and check every one against the triangle it stands for: centroid within eps_c, unit normal within
    eps_n, tight radius >= Rv + sigma'(MG) L, pn_eff >= pn (1 + KT / (L R)), band pn >= pn, L >= L, and the closed form
    pn (KQ L + KT / L) >= qn; the Morton-ordered copy is a permutation of the run's records that remembers each triangle's index."""
    monkeypatch.setenv("PT_TRICULL", "1")
    ps, _ = scenes.triangle_mesh_scene(n_triangles=3000, seed=77)
    n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
    abi.check(lib.pt_debug_flatten(C.byref(ps.desc), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
    blob = np.zeros((n_f4.value, 4), f32)
    FP = C.POINTER(C.c_float)
    abi.check(lib.pt_debug_flatten(C.byref(ps.desc), blob.ctypes.data_as(FP), len(blob), C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
    assert flags.value & 4
    bi = blob.view(np.int32)
    bu = blob.view(np.uint32).reshape(-1)
    runs = bi[:n_runs.value]
    tri = [r for r in runs if r[0] == 2][0]
    first, count = int(tri[1]), int(tri[2])
    assert bi[first - 1][0] == 1
    hdr = int(bi[first - 1][1])
    H = blob[hdr:hdr + 10]
    Hi = bi[hdr:hdr + 10]
    cell_first, cell_cand, tri_sorted, cell_q = [int(x) for x in Hi[4]]
    kq, p_per_L, kt = float(H[5][1]), float(H[5][2]), float(H[5][3])
    cq_lo, eps_c = H[7][:3].astype(np.float64), float(H[7][3])
    cq_step, eps_n = H[8][:3].astype(np.float64), float(H[8][3])
    k_loose, m_scale, cell_n = float(H[9][0]), float(H[9][1]), int(Hi[9][2])
    assert abs(m_scale - MG / M) < 1e-3 and 2.0 < k_loose < 2.5
    nx, ny, nz = [int(x) for x in Hi[1][:3]]
    ncell = nx * ny * nz
    cf = bu[4 * cell_first: 4 * cell_first + ncell + 1].astype(np.int64)
    total = int(cf[-1])
    pos = bu[4 * cell_cand: 4 * cell_cand + total].astype(np.int64)
    gq = bu[4 * cell_q: 4 * cell_q + 2 * total].reshape(-1, 2)
    gn = bu[4 * cell_n: 4 * cell_n + 2 * total].reshape(-1, 2)
    srt = blob[tri_sorted: tri_sorted + 3 * count].reshape(count, 3, 4)
    orig = srt[:, 2, 3].view(np.int32).astype(np.int64)
    assert sorted(orig.tolist()) == list(range(count))                      # a permutation ...
    recs = blob[first: first + 3 * count].reshape(count, 3, 4)
    assert np.array_equal(srt[:, :, :3], recs[orig][:, :, :3])               # ... of the run's own records (v0, edge1, edge2)
    for c in range(ncell):
        assert np.all(np.diff(pos[cf[c]:cf[c + 1]]) > 0)                      # a cell's candidates ascend in the Morton copy
    # the triangles, in float64
    v0 = recs[:, 0, :3].astype(np.float64); e1 = recs[:, 1, :3].astype(np.float64); e2 = recs[:, 2, :3].astype(np.float64)
    N = np.cross(e1, e2); nN = np.linalg.norm(N, axis=1)
    l1, l2 = np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1)
    L = np.maximum(l1, l2)
    cen = (v0 + (e1 + e2) / 3).astype(f32).astype(np.float64)                # the rounded centroid the host measures from
    rv = np.max(np.stack([np.linalg.norm(v0 - cen, axis=1), np.linalg.norm(v0 + e1 - cen, axis=1), np.linalg.norm(v0 + e2 - cen, axis=1)]), axis=0)
    centre = H[2][:3].astype(np.float64); R = float(H[2][3])
    pn = M * 17.5 * U * SAFE * L / nN
    qn = ((MA * 7 + 4) * U * l1 * l2 * SAFE + 2.0 ** -40) / nN
    sig_g = (6 / MG + 6 / MA + 1.2 / (MG - 1)) * L
    bf = lambda hi16: (hi16.astype(np.uint32) << 16).view(f32).astype(np.float64)
    s16 = lambda v: ((v.astype(np.int64) & 0xffff) ^ 0x8000) - 0x8000
    t = orig[pos]                                                            # the triangle of every grid entry
    lo32, st32 = H[7][:3].astype(f32), H[8][:3].astype(f32)
    dec = lambda kx, ky, kz: np.stack([(lo32[a] + (k.astype(f32) * st32[a]).astype(f32)).astype(f32) for a, k in enumerate((kx, ky, kz))], -1).astype(np.float64)  # the device's binary32 decode
    C_dec = dec(gq[:, 0] & 0xffff, gq[:, 0] >> 16, gq[:, 1] & 0xffff)
    dev = np.linalg.norm(C_dec - cen[t], axis=1)
    assert np.all(dev <= eps_c * (1 + 1e-6) + 1e-12), (dev.max(), eps_c, int(np.argmax(dev)), cq_step)
    assert np.all(bf(gq[:, 1] >> 16) >= rv[t] + sig_g[t] + eps_c)
    n_dec = np.stack([s16(gn[:, 0]), s16(gn[:, 0] >> 16), s16(gn[:, 1])], -1) / 32767.0
    assert np.all(np.linalg.norm(n_dec - N[t] / nN[t][:, None], axis=1) <= eps_n)
    assert np.all(bf(gn[:, 1] >> 16) >= pn[t] * (1 + kt / (L[t] * R)))
    # band records of the three levels (both orientations)
    n_band = 0
    for lv in range(3):
        T = bi[hdr + 10 + 3 * lv: hdr + 13 + 3 * lv]
        Rl = int(T[0][0])
        for o in (1, 2):
            tf, tc, tr = [int(x) for x in T[o][:3]]
            nc = 3 * Rl * Rl
            fr = bu[4 * tf: 4 * tf + nc + 1].astype(np.int64)
            k = int(fr[-1])
            idx = orig[bu[4 * tc: 4 * tc + k].astype(np.int64)]     # (listed by position in the Morton copy, like the grid's)
            q = bu[4 * tr: 4 * tr + 4 * k].reshape(-1, 4)
            nd = np.stack([s16(q[:, 0]), s16(q[:, 0] >> 16), s16(q[:, 1])], -1) / 32767.0
            assert np.all(np.linalg.norm(nd - N[idx] / nN[idx][:, None], axis=1) <= eps_n)
            pq, Lq = bf(q[:, 1] >> 16), bf(q[:, 3] >> 16)
            assert np.all(pq >= pn[idx]) and np.all(pq <= pn[idx] * 1.01) and np.all(Lq >= L[idx]) and np.all(Lq <= L[idx] * 1.01)
            assert np.all(pq * (kq * Lq + kt / Lq) >= qn[idx])
            Cd = dec(q[:, 2] & 0xffff, q[:, 2] >> 16, q[:, 3] & 0xffff)
            assert np.all(np.linalg.norm(Cd - cen[idx], axis=1) <= eps_c * (1 + 1e-6) + 1e-12)
            n_band += k
    assert total > 3000 and n_band > 3000


def test_integer_band_test_is_conservative():
    """Stage 1 of the band filter (pt_device.hpp: band_stage1) evaluates |d^ . n^| in integers: n^ and the ray's unit direction are
    rounded to k / 32767 per component and S = kn . kd is exact (two v_dot2_i32_i16).  Whenever the real band condition
    |d^ . n^| <= thr holds, the device's comparison  |S| <= (thr + eps_n + eps_d) * 32767^2 (1 + 2e-5)  must hold — emulated here in
    binary32 for random and for barely-inside pairs, with thr across the range of the three levels; and |S| stays below 2^31."""
    rng = np.random.default_rng(11)
    n_pairs = 400_000
    n = rng.normal(size=(n_pairs, 3)); n /= np.linalg.norm(n, axis=1)[:, None]
    # directions close to the band's edge: d = cos(a) t + sin(a) n with sin(a) = thr (1 - tiny) and thr log-uniform in [1e-5, 0.3]
    thr = 10 ** rng.uniform(-5, np.log10(0.3), n_pairs)
    t = np.cross(n, rng.normal(size=(n_pairs, 3))); t /= np.linalg.norm(t, axis=1)[:, None]
    inside = rng.uniform(0, 1, n_pairs) ** 0.05                      # mostly within a few per cent of the edge, on the inside
    sa = thr * inside * rng.choice([-1.0, 1.0], n_pairs)
    d = t * np.sqrt(1 - sa * sa)[:, None] + n * sa[:, None]
    assert np.all(np.abs(np.einsum("ij,ij->i", d, n)) <= thr * (1 + 1e-12))
    scale = 10 ** rng.uniform(-3, 3, n_pairs)                        # the ray's direction is not normalised
    d32 = (d * scale[:, None]).astype(f32)
    ua = dot32(d32, d32)
    dh = ((f32(1.0) / np.sqrt(ua)).astype(f32)[:, None] * d32).astype(f32)   # rsq(ua) * d: within a few ulp of the unit vector
    kd = np.rint(dh * f32(32767.0)).astype(np.int64)
    kn = np.rint(n * 32767.0).astype(np.int64)
    assert np.abs(kd).max() <= 32767 and np.abs(kn).max() <= 32767
    S = np.einsum("ij,ij->i", kn, kd)
    assert np.abs(S).max() < 2 ** 31 and np.abs(kn[:, :2] * kd[:, :2]).sum(1).max() < 2 ** 31
    eps_n = np.sqrt(3) * 0.5 / 32767 * (1 + 1e-6) + 3e-6              # what build_tri_pool measures, at its worst
    e1s = f32(eps_n) + f32(2.75e-5)
    lhs = np.abs(S).astype(f32)
    rhs = ((thr.astype(f32) + e1s).astype(f32) * f32(1.0737e9)).astype(f32)
    assert np.all(lhs <= rhs), int(np.argmax(lhs - rhs))
    # and it is a filter: of random pairs against a band of 1e-3 it rejects all but the ones near the band
    kd2 = rng.permutation(kd)
    S2 = np.abs(np.einsum("ij,ij->i", kn, kd2)).astype(f32)
    keep = S2 <= ((f32(1e-3) + e1s) * f32(1.0737e9)).astype(f32)
    assert keep.mean() < 0.002
