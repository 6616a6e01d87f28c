"""The inequalities the triangle pool's exactness rests on (path_tracer_amd/csrc/pt_tripool.hpp), checked on a binary32 emulation
of the reference's Moller-Trumbore test (triangle.hpp:58-100: same operations, same order, no contraction) for random rays and
for adversarial rays that lie almost in a triangle's plane:

  every (ray, triangle) pair the emulated test accepts is either GRAZING by the pool's own band test
  (|d . N'_i| < |d| (rho P_i + Q_i): it is then listed in the bin of the ray's direction in the direction map of its rho class), or
  NOT grazing and then
  (a) the exact line-plane point, clamped onto the forward ray, lies within sigma'_i = sigma_b,i + sigma_t,i of the triangle (so inside
      its box grown by sigma'_i) AND within sigma_t,i = 0.56 min(|e1|, |e2|) / (1.4 M - 1) of its plane (so the grid cell that contains that point lists the
      triangle; round 6 tightened both: the joint bound on the two barycentrics' errors, the plane slab by the ray-parameter term alone), and
  (b) the computed t is within kappa (relative) + 1.2 L_i / ((M - 1) |d|) of the exact parameter (so the walk's range
      [0, closest (1 + kappa)] reaches that cell);
  and in every case the ray's LINE passes within the radius the noise filter allows of the centroid.

The tables themselves (pt_debug_flatten / pt_debug_flatten_pool: what pt_scene_create uploads) are then checked against these
definitions — the grid's cell lists, the direction maps' bins, the compressed band records.  The GPU suite checks the walked structure
end to end against the oracle, bit for bit (tests/test_gpu_fuzz.py)."""
import ctypes as C

import numpy as np
import pytest

from path_tracer_amd import abi, scenes
from path_tracer_amd.scene import hittable_dtype

f32 = np.float32
U = 2.0 ** -24
M, MA, SAFE = 6.0, 256.0, 1.5   # pt_tripool.hpp: TriPoolTuning defaults and SAFE
MG, MAG = 1.4 * M, 1.4 * MA                # what the grazing threshold (with its SAFE) gives the grid's side (pt_tripool.hpp header)
BS, BT = 1 / MG + 2 / MAG, 1 / MG + 1 / MAG


def sigmas(l1, l2):
    """(sigma_t,i, sigma'_i): P' from P^ along the ray (the plane slab's half thickness); P' from the triangle (the grown box, the
    point-triangle distance)."""
    sig_t = 0.56 * np.minimum(l1, l2) / (MG - 1)
    return sig_t, (BS + 2 * U) * np.maximum(l1, l2) + BT * (l1 + l2) + sig_t


def cross32(a, b):
    return np.stack([(a[..., 1] * b[..., 2]).astype(f32) - (a[..., 2] * b[..., 1]).astype(f32),
                     (a[..., 2] * b[..., 0]).astype(f32) - (a[..., 0] * b[..., 2]).astype(f32),
                     (a[..., 0] * b[..., 1]).astype(f32) - (a[..., 1] * b[..., 0]).astype(f32)], -1).astype(f32)


def dot32(a, b):
    return (((a[..., 0] * b[..., 0]).astype(f32) + (a[..., 1] * b[..., 1]).astype(f32)).astype(f32) + (a[..., 2] * b[..., 2]).astype(f32)).astype(f32)


class Pool:
    """The pool's header and tables, parsed back out of what the flattener produces (host-only)."""

    def __init__(self, lib, ps, tuning=None):
        n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
        FP = C.POINTER(C.c_float)
        tp = C.byref(tuning) if tuning is not None else None
        abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), tp, None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
        blob = np.zeros((n_f4.value, 4), f32)
        abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), tp, blob.ctypes.data_as(FP), len(blob), C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
        assert flags.value & 4, "no triangle pool"
        n_pool = C.c_int64()
        abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), tp, None, 0, C.byref(n_pool)), "pt_debug_flatten_pool")
        pool = np.zeros((n_pool.value, 4), f32)
        abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), tp, pool.ctypes.data_as(FP), len(pool), C.byref(n_pool)), "pt_debug_flatten_pool")
        self.blob, self.pool = blob, pool
        bi = blob.view(np.int32)
        self.pu = pool.view(np.uint32).reshape(-1)
        tri = [r for r in bi[:n_runs.value] if r[0] == 2][0]
        self.first, self.count = int(tri[1]), int(tri[2])
        assert bi[self.first - 1][0] == 1
        hdr = int(bi[self.first - 1][1])
        self.H, self.Hi = blob[hdr:hdr + 12], bi[hdr:hdr + 12]
        H, Hi = self.H, self.Hi
        self.origin, self.inv_cell = H[0][:3].astype(np.float64), float(H[0][3])
        self.n = [int(x) for x in Hi[1][:3]]
        self.cell = float(H[1][3])
        self.centre, self.R = H[2][:3].astype(np.float64), float(H[2][3])
        self.rlimit2, self.kappa, self.n_tri, self.n_maps = float(H[3][0]), float(H[3][1]), int(Hi[3][2]), int(Hi[3][3])
        self.cell_first, self.cell_cand, self.tri_sorted, self.band = [int(x) & 0xffffffff for x in Hi[4]]
        self.kq, self.p_per_L, self.kt = float(H[5][1]), float(H[5][2]), float(H[5][3])
        self.cq_lo, self.eps_c = H[7][:3], float(H[7][3])
        self.cq_step, self.eps_n = H[8][:3], float(H[8][3])
        self.maps = [dict(R=int(Hi[9 + k][0]), rho_max=float(H[9 + k][1]), first=int(Hi[9 + k][2]) & 0xffffffff, cand=int(Hi[9 + k][3]) & 0xffffffff)
                     for k in range(self.n_maps)]
        srt = pool[self.tri_sorted: self.tri_sorted + 3 * self.count].reshape(self.count, 3, 4)
        self.sorted_recs = srt
        self.orig = srt[:, 2, 3].view(np.int32).astype(np.int64)            # Morton position -> triangle index in the run
        self.pos_of = np.empty(self.count, np.int64)
        self.pos_of[self.orig] = np.arange(self.count)
        ncell = self.n[0] * self.n[1] * self.n[2]
        self.cf = self.pu[4 * self.cell_first: 4 * self.cell_first + ncell + 1].astype(np.int64)
        raw = self.pu[4 * self.cell_cand: 4 * self.cell_cand + int(self.cf[-1])].astype(np.int64)
        self.cc = raw & 0x3ffffff        # position in the Morton-ordered copy
        self.cc_bits = raw >> 26         # which of the six face neighbours list the triangle too (-x +x -y +y -z +z)

    def cell_list(self, ix, iy, iz):
        c = (iz * self.n[1] + iy) * self.n[0] + ix
        return self.cc[self.cf[c]:self.cf[c + 1]]

    def bin_list(self, k, d):
        """The device's bin of direction d (binary32, as tri_pool_scan computes it) in map k: Morton positions listed there."""
        m = self.maps[k]
        R = m["R"]
        d = d.astype(f32)
        ad = np.abs(d)
        face = 0 if (ad[0] >= ad[1] and ad[0] >= ad[2]) else (1 if ad[1] >= ad[2] else 2)
        dk, da, db = d[face], d[(face + 1) % 3], d[(face + 2) % 3]
        rk = f32(1.0) / dk
        ci = int(min(max(np.floor((f32(da * rk) + f32(1.0)) * f32(0.5 * R)), 0), R - 1))
        cj = int(min(max(np.floor((f32(db * rk) + f32(1.0)) * f32(0.5 * R)), 0), R - 1))
        b = (face * R + cj) * R + ci
        fr = self.pu[4 * m["first"] + b: 4 * m["first"] + b + 2].astype(np.int64)
        return self.pu[4 * m["cand"] + fr[0]: 4 * m["cand"] + fr[1]].astype(np.int64)


def T_dist2(p, a, e1, e2):
    """Squared distance from the point p to the triangle (a, a + e1, a + e2): the closest-point regions (vertex / edge / face), binary64."""
    b, c = a + e1, a + e2
    ap = p - a
    d1, d2 = e1 @ ap, e2 @ ap
    if d1 <= 0 and d2 <= 0:
        return ap @ ap
    bp = p - b
    d3, d4 = e1 @ bp, e2 @ bp
    if d3 >= 0 and d4 <= d3:
        return bp @ bp
    vc = d1 * d4 - d3 * d2
    if vc <= 0 and d1 >= 0 and d3 <= 0:
        q = a + e1 * (d1 / (d1 - d3))
        return (p - q) @ (p - q)
    cp = p - c
    d5, d6 = e1 @ cp, e2 @ cp
    if d6 >= 0 and d5 <= d6:
        return cp @ cp
    vb = d5 * d2 - d1 * d6
    if vb <= 0 and d2 >= 0 and d6 <= 0:
        q = a + e2 * (d2 / (d2 - d6))
        return (p - q) @ (p - q)
    va = d3 * d6 - d5 * d4
    if va <= 0 and (d4 - d3) >= 0 and (d5 - d6) >= 0:
        q = b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))
        return (p - q) @ (p - q)
    den = va + vb + vc
    q = a + e1 * (vb / den) + e2 * (vc / den)
    return (p - q) @ (p - q)


def mesh_arrays(ps):
    h = np.frombuffer(ps.hittables, dtype=hittable_dtype)
    f = h["f"][1:-1].astype(f32)
    v0 = f[:, 0:3]
    e1, e2 = (f[:, 3:6] - v0).astype(f32), (f[:, 6:9] - v0).astype(f32)
    return v0, e1, e2


def test_accepted_pairs_are_band_or_grid_candidates(lib):
    ps, _ = scenes.triangle_mesh_scene(n_triangles=20_000)
    pool = Pool(lib, ps)
    assert pool.n_maps == 3 and pool.count == 20_000
    v0, e1, e2 = mesh_arrays(ps)
    e1d, e2d, v0d = e1.astype(np.float64), e2.astype(np.float64), v0.astype(np.float64)
    Nd = np.cross(e1d, e2d)
    nN = np.linalg.norm(Nd, axis=1)
    nh_all = Nd / nN[:, None]
    l1, l2 = np.linalg.norm(e1d, axis=1), np.linalg.norm(e2d, axis=1)
    L = np.maximum(l1, l2)
    P = M * 17.5 * U * L * SAFE
    Q = (MA * 7 + 4) * U * l1 * l2 * SAFE + 2.0 ** -40
    sig_t, sig = sigmas(l1, l2)
    centre = v0d.mean(0)
    R = np.linalg.norm(v0d - centre, axis=1).max()
    assert abs(R / pool.R - 1) < 1e-5 and np.allclose(centre, pool.centre, atol=1e-5)
    lo = np.minimum(np.minimum(v0d, v0d + e1d), v0d + e2d)
    hi = np.maximum(np.maximum(v0d, v0d + e1d), v0d + e2d)
    cen = v0d + (e1d + e2d) / 3
    Np = Nd.astype(f32).astype(np.float64)
    kr = 6 * SAFE * U * L * L * (17.5 + 7 * L / R)
    rng = np.random.default_rng(5)
    stats = dict(accepted=0, band=0, grid=0, far=0)

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
        lists = {}
        for i in np.nonzero(ok)[0]:
            stats["accepted"] += 1
            # distance of the ray's line from the centroid
            dist_line = np.linalg.norm(np.cross(cen[i] - od, dd)) / dn
            if band[i]:
                stats["band"] += 1
                a1 = ap[i] - 4 * U * L[i] * L[i] * dn
                if a1 > 0:
                    assert dist_line <= L[i] + kr[i] * rho * dn / a1, ("noise-radius filter", i)
                # ... and the pair is in the bin of this direction, in the map of the ray's rho class
                k = next((k for k, m in enumerate(pool.maps) if rho * 1.000003 <= m["rho_max"]), None)
                if k is None:
                    stats["far"] += 1      # beyond the last class: the device streams every band record
                    continue
                if k not in lists:
                    lists[k] = set(pool.bin_list(k, d).tolist())
                assert int(pool.pos_of[i]) in lists[k], ("direction map", k, i)
                continue
            stats["grid"] += 1
            a_ = -(dd @ Nd[i])
            th = (e2d[i] @ np.cross(od - v0d[i], e1d[i])) / a_
            Pp = od + max(th, 0.0) * dd
            assert np.max(np.maximum(np.maximum(lo[i] - Pp, Pp - hi[i]), 0)) <= sig[i], ("grown box", i)
            assert abs(nh_all[i] @ (Pp - v0d[i])) <= sig_t[i], ("plane slab", i)
            assert np.sqrt(T_dist2(Pp, v0d[i], e1d[i], e2d[i])) <= sig[i], ("point-triangle distance", i)
            stats["slab_used"] = max(stats.get("slab_used", 0.0), float(abs(nh_all[i] @ (Pp - v0d[i])) / sig_t[i]))
            stats["dist_used"] = max(stats.get("dist_used", 0.0), float(np.sqrt(T_dist2(Pp, v0d[i], e1d[i], e2d[i])) / sig[i]))
            assert abs(float(t[i]) - th) <= 2.2 / (MA - 1) * abs(th) + 1.2 * L[i] / ((M - 1) * dn) + 1e-12, ("t", i)
            # the cell that contains P' lists the triangle
            cxyz = np.floor((Pp - pool.origin) * pool.inv_cell).astype(int)
            if np.all(cxyz >= 0) and np.all(cxyz < pool.n):
                assert int(pool.pos_of[i]) in pool.cell_list(*cxyz), ("grid cell", i)

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
    for _ in range(60):    # ... and from far away (the outer rho classes and beyond)
        i = int(rng.integers(n))
        nh = Nd[i] / max(np.linalg.norm(Nd[i]), 1e-300)
        t1 = e1d[i] / l1[i]
        d = (t1 + nh * 10 ** rng.uniform(-6, -2)) * rng.uniform(0.3, 2)
        check(cen[i] - d / np.linalg.norm(d) * rng.uniform(15, 90), d)
    print("accepted pairs:", stats)   # (-s: how much of the two slacks the worst observed pair used)
    assert stats["accepted"] > 3000 and stats["band"] > 20 and stats["grid"] > 2000, stats


def test_grid_lists_every_cell_the_bound_allows(lib, monkeypatch):
    """The grid's tables against the REGION the bound speaks of, not only against the points rounding happens to reach: the exact line-plane
    point of an accepted pair that is not grazing lies in the enlarged triangle T+ (beta >= -T, gamma >= -T, beta + gamma <= 1 + u + S) and
    the point of the walked segment within sigma_t of it — every cell that contains such a point must list the triangle (corners and edges
    of T+ included, offsets of the full sigma_t in every direction)."""
    monkeypatch.setenv("PT_TRICULL", "1")
    ps, _ = scenes.triangle_mesh_scene(n_triangles=3000, seed=21)
    pool = Pool(lib, ps)
    v0, e1, e2 = mesh_arrays(ps)
    v0d, e1d, e2d = v0.astype(np.float64), e1.astype(np.float64), e2.astype(np.float64)
    l1, l2 = np.linalg.norm(e1d, axis=1), np.linalg.norm(e2d, axis=1)
    sig_t, _ = sigmas(l1, l2)
    rng = np.random.default_rng(9)
    top = 1 + U + BS
    checked = 0
    for _ in range(6000):
        i = int(rng.integers(len(v0)))
        kind = rng.integers(5)
        if kind == 0: b, g = -BT, -BT
        elif kind == 1: b, g = top + BT, -BT
        elif kind == 2: b, g = -BT, top + BT
        elif kind == 3:                                   # on an edge of T+
            w = rng.uniform()
            b, g = [(-BT + w * (top + 2 * BT), -BT), (-BT, -BT + w * (top + 2 * BT)), (-BT + w * (top + 2 * BT), top + BT - w * (top + 2 * BT))][int(rng.integers(3))]
        else:                                             # inside
            b = rng.uniform(-BT, top + BT); g = rng.uniform(-BT, top - b)
        ph = v0d[i] + b * e1d[i] + g * e2d[i]
        off = rng.normal(size=3)
        pp = ph + off / np.linalg.norm(off) * sig_t[i] * rng.choice([1.0, 1.0, rng.uniform()])
        cxyz = np.floor((pp - pool.origin) * pool.inv_cell).astype(int)
        if np.all(cxyz >= 0) and np.all(cxyz < pool.n):
            assert int(pool.pos_of[i]) in pool.cell_list(*cxyz), (i, kind, b, g)
            checked += 1
    assert checked > 5000


def test_direction_maps_list_every_triangle_a_direction_can_graze(lib):
    """The maps against their definition: for random directions (and directions on bin borders and face edges) every triangle whose
    REAL band test |d^ . n^_i| <= rho_max pn_i + qn_i can pass is listed in the device's bin of that direction — and the maps are not
    trivially full."""
    ps, _ = scenes.triangle_mesh_scene(n_triangles=6000, seed=3)
    pool = Pool(lib, ps)
    v0, e1, e2 = mesh_arrays(ps)
    e1d, e2d = e1.astype(np.float64), e2.astype(np.float64)
    Nd = np.cross(e1d, e2d)
    nN = np.linalg.norm(Nd, axis=1)
    nh = Nd / nN[:, None]
    l1, l2 = np.linalg.norm(e1d, axis=1), np.linalg.norm(e2d, axis=1)
    L = np.maximum(l1, l2)
    pn = M * 17.5 * U * SAFE * L / nN
    qn = ((MA * 7 + 4) * U * l1 * l2 * SAFE + 2.0 ** -40) / nN
    rng = np.random.default_rng(8)
    dirs = [rng.normal(size=3) for _ in range(300)]
    for k, m in enumerate(pool.maps):   # directions that sit on bin borders and on the cube's edges / corners
        R = m["R"]
        for _ in range(100):
            p, q = (rng.integers(0, R + 1, 2) * 2.0 / R - 1.0)
            face = int(rng.integers(3))
            d = np.zeros(3); d[face] = rng.choice([-1.0, 1.0]); d[(face + 1) % 3] = p * d[face]; d[(face + 2) % 3] = q * d[face]
            dirs.append(d * rng.uniform(0.1, 3))
    listed_share = []
    for k, m in enumerate(pool.maps):
        rho_max = m["rho_max"] * (1 + 2e-6)
        tau = rho_max * pn + qn
        for d in dirs:
            d32 = np.asarray(d, f32)
            dh = d32.astype(np.float64) / np.linalg.norm(d32.astype(np.float64))
            graze = np.nonzero(np.abs(nh @ dh) <= tau)[0]
            got = set(pool.bin_list(k, d32).tolist())
            missing = [int(i) for i in graze if int(pool.pos_of[i]) not in got]
            assert not missing, (k, d, missing[:5])
            listed_share.append(len(got) / pool.count)
    assert np.mean(listed_share) < 0.5


def test_pool_thresholds_and_tables(lib, monkeypatch):
    """Long triangle runs (>= 4096) get a pool by default — the 100 k-triangle mesh of BASELINE config 5 does — shorter ones only
    with PT_TRICULL=1 (>= 256: the fuzz fields), none with PT_NO_TRICULL; the tables live in a buffer of their own, the blob grows by
    the pool's header only."""
    ps, _ = scenes.triangle_mesh_scene()
    st = (C.c_int32 * 8)()
    abi.check(lib.pt_debug_tri_pool(C.byref(ps.desc), st), "pt_debug_tri_pool")
    assert st[0] == 100_000 and 0 <= st[1] < 2000
    assert st[2] > 10_000 and st[3] > 10_000 and [(st[4] >> 20) & 1023, (st[4] >> 10) & 1023, st[4] & 1023] == [128, 64, 32]   # the three direction maps (K entries: the first, the other two; resolutions since round 6)
    assert 5000 < st[5] < 100000 and st[6] * 16 < 5.0e6                                             # cells per triangle (x 1000); the blob stays the plain 4.8 MB
    n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
    abi.check(lib.pt_debug_flatten(C.byref(ps.desc), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten")
    assert flags.value & 4 and n_runs.value == 3
    n_pool = C.c_int64()
    abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), None, None, 0, C.byref(n_pool)), "pt_debug_flatten_pool")
    assert 1.0e8 < n_pool.value * 16 < 1.0e9  # (round 6: under a gigabyte)
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
        abi.check(lib.pt_debug_flatten_pool(C.byref(sc.desc), None, None, 0, C.byref(n_pool)), "pt_debug_flatten_pool")
        assert n_pool.value == 0


def test_large_meshes_keep_their_pool(lib):
    """rounds 3-4 kept the pool's tables in the blob, whose 24-bit record offsets are hit-id bits — a
    mesh beyond ~830 k triangles lost its pool (8x slower class), one beyond 5.5 M failed to flatten.  The tables now live in their own
    buffer (32-bit offsets of 16-byte records) and hit ids carry 25-bit offsets: a 2 M-triangle mesh is pooled (its direction maps
    within a small budget here, to keep the CPU suite quick), and the plain blob of a 6 M-triangle mesh fits."""
    ps, _ = scenes.triangle_mesh_scene(n_triangles=2_000_000)
    st = (C.c_int32 * 8)()
    t = abi.tuning(tri_budget_mb=64)
    n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
    abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), C.byref(t), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten_tuned")
    assert flags.value & 4                                        # tri_pooled
    assert 2_000_000 * 3 < n_f4.value < 2_000_000 * 3 + 64        # the blob: three records per triangle + headers
    n_pool = C.c_int64()
    abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), C.byref(t), None, 0, C.byref(n_pool)), "pt_debug_flatten_pool")
    assert n_pool.value > 2_000_000 * 4
    del ps
    big, _ = scenes.triangle_mesh_scene(n_triangles=6_000_000)
    t = abi.tuning(tri_pool=-1)
    abi.check(lib.pt_debug_flatten_tuned(C.byref(big.desc), C.byref(t), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags)), "pt_debug_flatten_tuned")
    assert (1 << 24) < 6_000_000 * 3 < n_f4.value < (1 << 25)


def test_compressed_band_records_round_to_the_safe_side(lib, monkeypatch):
    """The device gathers QUANTISED band records (pt_tripool.hpp "compressed records").  Parse them back out of the pool of a
    3000-triangle field and check every one against the triangle it stands for: centroid within eps_c, unit normal within eps_n,
    pn >= pn, L >= L, and the closed form pn (KQ L + KT / L) >= qn; the Morton-ordered copy is a permutation of the run's records that
    remembers each triangle's index; cell lists and bins ascend in it."""
    monkeypatch.setenv("PT_TRICULL", "1")
    ps, _ = scenes.triangle_mesh_scene(n_triangles=3000, seed=77)
    pool = Pool(lib, ps)
    count, orig, srt = pool.count, pool.orig, pool.sorted_recs
    assert sorted(orig.tolist()) == list(range(count))                      # a permutation ...
    recs = pool.blob[pool.first: pool.first + 3 * count].reshape(count, 3, 4)
    assert np.array_equal(srt[:, :, :3], recs[orig][:, :, :3])               # ... of the run's own records (v0, edge1, edge2)
    ncell = pool.n[0] * pool.n[1] * pool.n[2]
    for c in range(ncell):
        assert np.all(np.diff(pool.cc[pool.cf[c]:pool.cf[c + 1]]) > 0)        # a cell's candidates ascend in the Morton copy
    # the neighbour bits of every entry against the lists themselves
    nx, ny, nz = pool.n
    sets = [set(pool.cc[pool.cf[c]:pool.cf[c + 1]].tolist()) for c in range(ncell)]
    rng = np.random.default_rng(4)
    for c in rng.integers(0, ncell, 300):
        x, y, z = int(c % nx), int((c // nx) % ny), int(c // (nx * ny))
        nbr = [(x > 0, c - 1), (x + 1 < nx, c + 1), (y > 0, c - nx), (y + 1 < ny, c + nx), (z > 0, c - nx * ny), (z + 1 < nz, c + nx * ny)]
        for pos, bits in zip(pool.cc[pool.cf[c]:pool.cf[c + 1]], pool.cc_bits[pool.cf[c]:pool.cf[c + 1]]):
            want = sum((1 << k) for k, (ok, c2) in enumerate(nbr) if ok and int(pos) in sets[int(c2)])
            assert int(bits) == want, (c, int(pos), int(bits), want)
    for m in pool.maps:
        nb = 3 * m["R"] * m["R"]
        fr = pool.pu[4 * m["first"]: 4 * m["first"] + nb + 1].astype(np.int64)
        cand = pool.pu[4 * m["cand"]: 4 * m["cand"] + int(fr[-1])].astype(np.int64)
        assert np.all(np.diff(fr) >= 0) and cand.max() < count
        for b in np.random.default_rng(1).integers(0, nb, 400):
            assert np.all(np.diff(cand[fr[b]:fr[b + 1]]) > 0)                 # a bin's candidates too
    # the triangles, in float64
    v0 = recs[:, 0, :3].astype(np.float64); e1 = recs[:, 1, :3].astype(np.float64); e2 = recs[:, 2, :3].astype(np.float64)
    N = np.cross(e1, e2); nN = np.linalg.norm(N, axis=1)
    l1, l2 = np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1)
    L = np.maximum(l1, l2)
    cen = (v0 + (e1 + e2) / 3).astype(f32).astype(np.float64)                # the rounded centroid the host measures from
    pn = M * 17.5 * U * SAFE * L / nN
    qn = ((MA * 7 + 4) * U * l1 * l2 * SAFE + 2.0 ** -40) / nN
    bf = lambda hi16: (hi16.astype(np.uint32) << 16).view(f32).astype(np.float64)
    s16 = lambda v: ((v.astype(np.int64) & 0xffff) ^ 0x8000) - 0x8000
    lo32, st32 = pool.cq_lo.astype(f32), pool.cq_step.astype(f32)
    dec = lambda kx, ky, kz: np.stack([(lo32[a] + (k.astype(f32) * st32[a]).astype(f32)).astype(f32) for a, k in enumerate((kx, ky, kz))], -1).astype(np.float64)  # the device's binary32 decode
    q = pool.pu[4 * pool.band: 4 * pool.band + 4 * count].reshape(-1, 4)     # in Morton order
    idx = orig
    nd = np.stack([s16(q[:, 0]), s16(q[:, 0] >> 16), s16(q[:, 1])], -1) / 32767.0
    assert np.all(np.linalg.norm(nd - N[idx] / nN[idx][:, None], axis=1) <= pool.eps_n)
    pq, Lq = bf(q[:, 1] >> 16), bf(q[:, 3] >> 16)
    assert np.all(pq >= pn[idx]) and np.all(pq <= pn[idx] * 1.01) and np.all(Lq >= L[idx]) and np.all(Lq <= L[idx] * 1.01)
    assert np.all(pq * (pool.kq * Lq + pool.kt / Lq) >= qn[idx])
    Cd = dec(q[:, 2] & 0xffff, q[:, 2] >> 16, q[:, 3] & 0xffff)
    assert np.all(np.linalg.norm(Cd - cen[idx], axis=1) <= pool.eps_c * (1 + 1e-6) + 1e-12)
    assert int(pool.cf[-1]) > 3000


def test_triangles_without_a_normal_and_dead_ones(lib, monkeypatch):
    """Edges that are parallel (|N| = 0) leave no direction to index: such a triangle is listed in every bin with the record that passes
    every filter; a triangle with an edge of zero length can never be accepted (a = +-0 for every ray) and is in no table."""
    monkeypatch.setenv("PT_TRICULL", "1")
    ps, _ = scenes.triangle_mesh_scene(n_triangles=400, seed=5)
    h = np.frombuffer(ps.hittables, dtype=hittable_dtype)
    f = h["f"]
    f[10, 3:6] = f[10, 0:3] + f32([0.1, 0.0, 0.0]); f[10, 6:9] = f[10, 0:3] + f32([0.25, 0.0, 0.0])     # parallel edges
    f[11, 3:6] = f[11, 0:3]                                                                              # a zero edge
    pool = Pool(lib, ps)
    p10, p11 = int(pool.pos_of[9]), int(pool.pos_of[10])                     # (hittable 0 is the ground sphere: run index = hittable - 1)
    q = pool.pu[4 * pool.band: 4 * pool.band + 4 * pool.count].reshape(-1, 4)
    assert (q[p10, 1] >> 16) == 0x7f7f and (q[p10, 0] == 0)
    assert np.all(q[p11] == 0)
    rng = np.random.default_rng(2)
    for _ in range(50):
        got = pool.bin_list(0, rng.normal(size=3).astype(f32))
        assert p10 in got and p11 not in got
    assert p11 not in pool.cc


def test_integer_band_test_is_conservative():
    """Stage 1 of the band filter (pt_device.hpp: band_stage1) evaluates |d^ . n^| in integers: n^ and the ray's unit direction are
    rounded to k / 32767 per component and S = kn . kd is exact (two v_dot2_i32_i16).  Whenever the real band condition
    |d^ . n^| <= thr holds, the device's comparison  |S| <= (thr + eps_n + eps_d) * 32767^2 (1 + 2e-5)  must hold — emulated here in
    binary32 for random and for barely-inside pairs, with thr across the range of band widths; and |S| stays below 2^31."""
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
