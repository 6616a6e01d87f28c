"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/pt_render.h declares, agrees with the header on struct sizes, flattens scenes as documented,
and reports malformed input through error codes (no compute calls: those need the GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import scenes_small as S
from path_tracer_amd import abi, scenes
from path_tracer_amd.scene import (box, camera, constant_medium, lambertian_material, metal_material, pack, sphere,
                                   triangle, xy_rect, xz_rect)

HEADER = Path(__file__).resolve().parent.parent / "include" / "pt_render.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libpt_render.so does not export {n}"
    assert set(names) == set(abi.SIGNATURES), "abi.py and pt_render.h disagree on the entry points"


def test_abi_version_and_errors(lib):
    assert lib.pt_abi_version() == abi.PT_ABI_VERSION
    assert lib.pt_error_string(abi.PT_OK) == b"ok"
    assert b"scene" in lib.pt_error_string(abi.PT_ERR_BAD_SCENE)


def test_struct_sizes_match_header():
    text = HEADER.read_text()
    for name, size in (("PtHittable", 64), ("PtMaterial", 32), ("PtTexture", 48), ("PtCamera", 96)):
        assert f"typedef struct {name}" in text
        assert C.sizeof(getattr(abi, name)) == size


def test_camera_matches_oracle(orc):
    for args in [((13, 3, 3), (0, -1, 0), (0, 1, 0), 40, 400 / 225, 0.04, 13.928, 0, 1),
                 ((278, 278, -800), (278, 278, 0), (0, 1, 0), 40, 16 / 9, 0, 800, 0, 1),
                 ((1, 2, 3), (-4, 0.5, 2), (0.1, 1, 0.2), 75.5, 1.25, 0.5, 3.3, 0.25, 0.75)]:
        a = camera(*args)
        b = orc.camera_init(*args)
        assert bytes(a.c) == bytes(b)


def test_framebuffer_sizes(lib):
    p = abi.PtRenderParams(1920, 1080, 1, 50, 0, 1, 0, 0)
    assert lib.pt_framebuffer_floats(C.byref(p)) == 1920 * 1080 * 3
    p = abi.PtRenderParams(1920, 1080, 1, 50, 3, 8, 0, 0)
    assert lib.pt_shard_tiles(C.byref(p)) == 240 * 135 // 8
    assert lib.pt_framebuffer_floats(C.byref(p)) == 240 * 135 // 8 * 64 * 3
    p = abi.PtRenderParams(21, 13, 1, 50, 0, 4, 0, 0)  # 3x2 tiles, padded to 2 per shard
    assert lib.pt_shard_tiles(C.byref(p)) == 2
    for bad in (abi.PtRenderParams(0, 8, 1, 1, 0, 1, 0, 0), abi.PtRenderParams(8, 8, 0, 1, 0, 1, 0, 0),
                abi.PtRenderParams(8, 8, 1, 1, 2, 2, 0, 0), abi.PtRenderParams(8, 8, 1, -1, 0, 1, 0, 0)):
        assert lib.pt_framebuffer_floats(C.byref(bad)) == -1


def flatten(lib, ps):
    n_f4, n_runs, flags = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.pt_debug_flatten(C.byref(ps.desc), None, 0, C.byref(n_f4), C.byref(n_runs), None, 0, C.byref(flags))
    if rc:
        return rc, None, None, None, None
    blob = np.zeros((n_f4.value, 4), np.float32)
    mats = np.zeros((ps.n_materials * 4, 4), np.float32)
    FP = C.POINTER(C.c_float)
    rc = lib.pt_debug_flatten(C.byref(ps.desc), blob.ctypes.data_as(FP), len(blob), None, None,
                              mats.ctypes.data_as(FP), len(mats), None)
    return rc, blob, mats, n_runs.value, flags.value


def test_flatten_preserves_list_order(lib):
    """Traversal order is semantics (constant_medium's in-traversal RNG draw, equal-t ties): runs are
    maximal stretches of one kind in list order, and record i of a run is hittable first+i."""
    m = lambertian_material((0.5, 0.5, 0.5))
    hs = [sphere((0, 0, 0), 1, m), sphere((1, 0, 0), 2, m), xy_rect(0, 1, 0, 1, 5, m), xz_rect(0, 1, 0, 1, 6, m),
          sphere((2, 0, 0), 3, m), box((0, 0, 0), (1, 1, 1), m), triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), m),
          triangle((0, 0, 1), (1, 0, 1), (0, 1, 1), m), constant_medium(sphere((0, 0, 0), 1, m), 2.0, (1, 1, 1)),
          sphere((3, 0, 0), 4, m)]
    ps = pack(hs)
    rc, blob, mats, n_runs, flags = flatten(lib, ps)
    assert rc == 0 and n_runs == 7 and flags == 2
    runs = blob[:n_runs].view(np.int32)
    assert runs[:, 0].tolist() == [0, 1, 0, 3, 2, 4, 0]       # device kinds: sphere, rect, sphere, box, tri, medium, sphere
    assert runs[:, 2].tolist() == [2, 2, 1, 1, 2, 1, 1]        # counts
    assert runs[:, 3].tolist() == [0, 2, 4, 5, 6, 8, 9]        # first hittable of each run
    sizes = {0: 3, 1: 2, 2: 3, 3: 2, 4: 4}
    off = n_runs
    for kind, first, count, _ in runs:
        if kind == 0:  # a sphere run is preceded by its static / moving offset lists (4 per F4) and an aux F4
            off += (int(count) + 3) // 4 + 1  # nothing moves here: one list; a handful of spheres: no grid
            aux = blob[first - 1]
            assert aux.view(np.int32)[2] == count and aux.view(np.int32)[3] == 1  # all static, uniform interval
            lst = blob[first - 1 - (int(count) + 3) // 4:first - 1].view(np.int32).reshape(-1)
            assert lst[:count].tolist() == [3 * i for i in range(count)] and (lst[count:] == 3 * (count - 1)).all()
        if kind in (1, 3):  # rect and box runs carry one aux F4 (and a stretch with enough boxes a slab pool table before it)
            aux = blob[first - 1].view(np.int32)
            assert aux[1] == 0 and aux[3] == 0   # two rects, then (after a sphere) a single box: no pool pays for itself here
            off += 1
        if kind == 2:  # a triangle run carries one aux F4: (has a triangle pool, header offset, -, -); two triangles: no pool
            aux = blob[first - 1].view(np.int32)
            assert aux[0] == 0 and aux[1] == 0
            off += 1
        assert first == off
        off += sizes[int(kind)] * int(count)
    assert off == len(blob)
    # sphere record: (c0, r^2) (r, mat, t0, t1) (c1 - c0, hittable index)
    s1 = blob[runs[0, 1] + 3: runs[0, 1] + 6]
    assert s1[0].tolist() == [1, 0, 0, 4] and s1[1, 0] == 2 and s1[2].view(np.int32)[3] == 1
    # rect record carries its axis (xy=0, xz=1)
    r = blob[runs[1, 1]: runs[1, 1] + 4]
    assert r[1].view(np.int32)[2] == 0 and r[3].view(np.int32)[2] == 1 and r[1, 0] == 5 and r[3, 0] == 6
    # triangle record stores edges v1-v0, v2-v0
    t = blob[runs[4, 1]: runs[4, 1] + 3]
    assert t[1, :3].tolist() == [1, 0, 0] and t[2, :3].tolist() == [0, 1, 0]
    # medium record: boundary kind, neg_inv_density = -1/2
    md = blob[runs[5, 1]]
    assert md.view(np.int32)[0] == 0 and md[1] == np.float32(-0.5)


def test_flatten_slab_pools(lib):
    """A maximal stretch of consecutive rect / box runs with at least three boxes on a fast_ok scene gets a slab pool table (slab
    entries, then exact entries carrying the hit id of the hittable's own record; a rect's exact entry has -inf on its own
    axis: pt_device.hpp slab_pool); a non-finite / huge coordinate anywhere switches the straight-line paths, and with them
    the pools, off."""
    m = lambertian_material((0.5, 0.5, 0.5))
    hs = [box((0, 0, 0), (1, 2, 3), m), box((-7, 0, 0), (1, 1, 1), m), xz_rect(0, 1, 2, 5.5, 4, m), box((0, 0, 0), (1, 1, 5), m),
          sphere((0, 0, 0), 1, m), box((0, 0, 0), (9, 1, 1), m)]
    rc, blob, mats, n_runs, flags = flatten(lib, pack(hs))
    assert rc == 0 and n_runs == 5
    runs = blob[:n_runs].view(np.int32)
    assert runs[:, 0].tolist() == [3, 1, 3, 0, 3]
    aux = blob[runs[0, 1] - 1]
    assert aux[0] == 7.0 and aux.view(np.int32)[1:].tolist() == [3, n_runs, 4]      # spans three runs, four entries, table first
    pool = blob[n_runs:n_runs + 16]
    assert pool[4].tolist()[:3] == [0, 4, 2] and pool[5].tolist()[:3] == [1, 4, 5.5]    # xz_rect: x, the plane y = 4, z
    assert pool[8 + 4].tolist()[:3] == [0, 4, 2] and pool[8 + 5, 1] == -np.inf and pool[8 + 5, 0] == 1 and pool[8 + 5, 2] == 5.5
    ids = pool[8::2].view(np.int32)[:, 3]
    assert (ids >> 28).tolist() == [3, 3, 1, 3]              # hit_pack: kind at bit 28, 25-bit record offsets (round 5)
    assert (ids & 0x1ffffff).tolist() == [runs[0, 1], runs[0, 1] + 2, runs[1, 1], runs[2, 1]]
    for r in (1, 2):
        assert blob[runs[r, 1] - 1].view(np.int32)[1] == 0                                # members, not heads
    assert blob[runs[4, 1] - 1].view(np.int32)[1] == 0                                    # a single box: not worth it
    hs[1] = box((-7, 0, 0), (1, 1, 3e18), m)
    rc, blob, mats, n_runs, flags = flatten(lib, pack(hs))
    assert rc == 0 and blob[blob[:n_runs].view(np.int32)[0, 1] - 1].view(np.int32)[1] == 0


def decode_sphere_aux(blob, first, count):
    """The records in front of a sphere run (pt_flatten.hpp: put_sphere_run_aux), decoded: dict with the aux fields, the
    full static / moving offset lists and, when the run has one, the grid (dims, cell table, candidates, big lists)."""
    aux = blob[first - 1]
    flags, ns = int(aux.view(np.int32)[3]), int(aux.view(np.int32)[2])
    n_foreign, flags = flags >> 8, flags & 255  # (round 6: the static list also names the absorbed spheres of later runs; aux.z counts them in)
    qs, qm = (ns + 3) // 4, (count - (ns - n_foreign) + 3) // 4
    end = first - 1 - (4 if flags & 4 else 0)
    out = dict(t0=float(aux[0]), t1=float(aux[1]), ns=ns, flags=flags, n_foreign=n_foreign,
               static=blob[end - qs - qm:end - qm].view(np.int32).reshape(-1), moving=blob[end - qm:end].view(np.int32).reshape(-1))
    if flags & 4:
        g0, g1, g2, g3 = blob[first - 5], blob[first - 4], blob[first - 3], blob[first - 2]
        n_cell, n_cand, qbs, qbm = (int(v) for v in g3.view(np.int32))
        big = end - qs - qm
        cand0 = big - qbs - qbm - n_cand
        out.update(origin=g0[:3].copy(), inv_cell=float(g0[3]), dims=[int(v) for v in g1.view(np.int32)[:3]], cell=float(g1[3]),
                   center=g2[:3].copy(), rlimit2=float(g2[3]),
                   cells=blob[cand0 - n_cell:cand0].view(np.uint32).reshape(-1), cand=blob[cand0:cand0 + n_cand].view(np.uint16).reshape(-1),
                   big_static=blob[big - qbs - qbm:big - qbm].view(np.int32).reshape(-1), big_moving=blob[big - qbm:big].view(np.int32).reshape(-1),
                   start=cand0 - n_cell)
    else:
        out["start"] = end - qs - qm
    return out


def test_flatten_sphere_run_lists(lib):
    """pt_flatten.hpp put_sphere_run_aux: the static and the moving spheres of a run as two lists of record offsets (list
    order, padded to a multiple of four by repeating the last entry); aux = (time0, time1, number of static spheres,
    flags: 1 = one shutter interval for all moving spheres, 2 = something moves, 4 = the run has a culling grid)."""
    m = lambertian_material((0.5, 0.5, 0.5))
    def scene(intervals, n=40):
        hs = []
        for i in range(n):
            if i in intervals:
                t0, t1 = intervals[i]
                hs.append(sphere((i, 0, 0), (i, 1, 0), t0, t1, 0.5, m))
            else:
                hs.append(sphere((i, 0, 0), 0.5, m))
        return pack(hs)
    moving = {3: (0.0, 1.0), 31: (0.0, 1.0), 32: (0.0, 1.0), 39: (0.0, 1.0)}
    rc, blob, mats, n_runs, flags = flatten(lib, scene(moving))
    assert rc == 0 and n_runs == 1
    first = int(blob[0].view(np.int32)[1])
    d = decode_sphere_aux(blob, first, 40)
    assert d["flags"] == 3 and (d["t0"], d["t1"], d["ns"]) == (0.0, 1.0, 36)  # fewer than 48 spheres: no grid
    assert d["start"] == 1  # right behind the run header
    assert d["static"][:36].tolist() == [3 * i for i in range(40) if i not in moving] and (d["static"][36:] == d["static"][35]).all()
    assert d["moving"].tolist() == [3 * i for i in sorted(moving)]
    for i in range(40):  # the per-record flag (sign of r^2) agrees with the lists
        assert (blob[first + 3 * i, 3] < 0) == (i in moving)
    moving[17] = (0.25, 1.0)  # a second shutter interval: the run falls back to the one-sphere-at-a-time scan
    rc, blob, *_ = flatten(lib, scene(moving))
    assert blob[int(blob[0].view(np.int32)[1]) - 1].view(np.int32)[3] == 2


def test_flatten_sphere_grid(lib):
    """pt_flatten.hpp build_sphere_grid: a run of >= 48 small spheres gets a uniform grid; every small sphere is listed in
    every cell its box [centre range -+ (|r| + margin)] touches, large spheres go to the "big" lists, and the full lists
    still name every sphere (the fallback scan)."""
    ps, _ = scenes.build("smoke", textures="procedural")
    rc, blob, mats, n_runs, flags = flatten(lib, ps)
    assert rc == 0
    runs = blob[:n_runs].view(np.int32)
    kind, first, count, first_h = (int(v) for v in runs[0])
    assert kind == 0 and count == 483 and first_h == 0
    d = decode_sphere_aux(blob, first, count)
    assert d["flags"] == 7
    # round 6, absorbed sphere runs: the glowing ball (run 2) and the five big spheres (run 4) sit behind the pyramid / the rect — kinds that
    # accept t == max — so run 0's lists test them too and the resident kernels skip their runs (kind | 8 in the run header)
    assert runs[:, 0].tolist() == [0, 2, 0 | 8, 1, 0 | 8, 3, 4] and d["n_foreign"] == 6
    own = {3 * i for i in range(count)}
    foreign = {int(runs[2, 1]) - first, *(int(runs[4, 1]) - first + 3 * i for i in range(5))}
    assert set(d["static"].tolist()) - own == foreign and set(d["big_static"].tolist()) - own == foreign
    n_f4, n_r = C.c_int32(), C.c_int32()
    t_off = abi.tuning(sphere_merge=-1)
    abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), C.byref(t_off), None, 0, C.byref(n_f4), C.byref(n_r), None, 0, None), "pt_debug_flatten_tuned")
    plain = np.zeros((n_f4.value, 4), np.float32)
    abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), C.byref(t_off), plain.ctypes.data_as(C.POINTER(C.c_float)), len(plain), None, None, None, 0, None), "pt_debug_flatten_tuned")
    assert plain[:n_runs].view(np.int32)[:, 0].tolist() == [0, 2, 0, 1, 0, 3, 4]  # PtTuning.sphere_merge = -1: every run where it stands
    d["static"] = np.array([o for o in d["static"].tolist() if o in own], np.int32)
    d["big_static"] = np.array([o for o in d["big_static"].tolist() if o in own], np.int32)
    nx, ny, nz = d["dims"]
    assert len(d["cells"]) >= nx * ny * nz and ny == 1 and nx >= 10 and nz >= 10
    assert abs(d["cell"] * d["inv_cell"] - 1) < 1e-6 and abs(d["cell"] - 3.0 * (0.2 + 0.1)) < 1e-4   # margin 0.5 r, cell 3 (r + margin)
    assert d["big_static"][:1].tolist() == [0] and len(set(d["big_static"].tolist())) == 1 and len(d["big_moving"]) == 0  # the ground sphere
    listed = set()
    for c in range(nx * ny * nz):
        hdr = int(d["cells"][c])
        cnt, off = hdr & 255, hdr >> 8
        for e in d["cand"][off:off + cnt]:
            o = 3 * (int(e) & 0x7FFF)  # candidates: 16-bit sphere index in the run | moving << 15
            listed.add(o)
            assert 0 < o < 3 * count and ((int(e) >> 15) & 1) == int(blob[first + o, 3] < 0)  # moving bit = the record's flag
            # the cell really touches the sphere's inflated box
            i = o // 3
            f = np.array(ps.hittables[i].f[:9], np.float32)
            lo = np.minimum(f[0:3], f[3:6]) - (0.2 + 0.1 + 1e-3); hi = np.maximum(f[0:3], f[3:6]) + (0.2 + 0.1 + 1e-3)
            cx, cz = c % nx, c // (nx * ny)
            cell_lo = d["origin"] + np.array([cx, 0, cz]) * d["cell"]; cell_hi = cell_lo + d["cell"]
            assert (cell_lo[[0, 2]] <= hi[[0, 2]] + 1e-4).all() and (cell_hi[[0, 2]] >= lo[[0, 2]] - 1e-4).all()
    assert listed == {3 * i for i in range(1, count)}  # every small sphere sits in some cell
    assert set(d["static"].tolist()) | set(d["moving"].tolist()) == {3 * i for i in range(count)}
    assert d["rlimit2"] > 80.0 ** 2  # rays starting within > 80 units of the field may use the grid


def test_flatten_badouel_triangles_are_runs_of_their_own(lib):
    """PtHittable.strategy (triangle.hpp:102-103): a Badouel-strategy triangle gets device kind 5 — its own run, in list
    order, same record as a Moller-Trumbore triangle; an unknown strategy is a malformed scene."""
    m = lambertian_material((0.5, 0.5, 0.5))
    t = lambda z, st="moller_trumbore": triangle((0, 0, z), (1, 0, z), (0, 1, z), m, st)  # noqa: E731
    ps = pack([t(0), t(1, "badouel"), t(2, "badouel"), t(3), sphere((0, 0, 0), 1, m)])
    assert [ps.hittables[i].strategy for i in range(5)] == [0, 1, 1, 0, 0]
    rc, blob, mats, n_runs, flags = flatten(lib, ps)
    assert rc == 0 and n_runs == 4
    runs = blob[:n_runs].view(np.int32)
    assert runs[:, 0].tolist() == [2, 5, 2, 0] and runs[:, 2].tolist() == [1, 2, 1, 1] and runs[:, 3].tolist() == [0, 1, 3, 4]
    rec = blob[runs[1, 1]: runs[1, 1] + 3]
    assert rec[0, :3].tolist() == [0, 0, 1] and rec[1, :3].tolist() == [1, 0, 0] and rec[2, :3].tolist() == [0, 1, 0]
    ps.hittables[1].strategy = 7
    assert flatten(lib, ps)[0] == abi.PT_ERR_BAD_SCENE


def test_flatten_flags_and_materials(lib):
    ps, _ = S.mixed_scene()
    rc, blob, mats, n_runs, flags = flatten(lib, ps)
    assert rc == 0 and flags == 3  # image texture + medium
    ps, _ = S.cornell_scene()
    rc, blob, mats, n_runs, flags = flatten(lib, ps)
    assert rc == 0 and flags == 0 and n_runs == 3  # boxes, xy_rect, boxes
    kinds = mats[0::4].view(np.int32)[:, 0].tolist()
    assert kinds == [abi.PT_MAT_LAMBERTIAN, abi.PT_MAT_LAMBERTIAN, abi.PT_MAT_LIGHTSOURCE, abi.PT_MAT_LAMBERTIAN]
    assert mats[2 * 4 + 1, :3].tolist() == [15, 15, 15]  # light colour inlined from its solid texture


@pytest.mark.parametrize("mutate,code", [
    (lambda ps: setattr(ps.hittables[0], "kind", 9), abi.PT_ERR_BAD_SCENE),
    (lambda ps: setattr(ps.hittables[1], "material", -1), abi.PT_ERR_BAD_SCENE),
    (lambda ps: setattr(ps.materials[0], "kind", 7), abi.PT_ERR_BAD_SCENE),
    (lambda ps: setattr(ps.materials[0], "texture", 1000), abi.PT_ERR_BAD_SCENE),
    (lambda ps: setattr(ps.textures[0], "kind", 3), abi.PT_ERR_BAD_SCENE),
])
def test_malformed_scene_is_an_error_code(lib, mutate, code):
    ps, _ = S.cornell_scene()
    mutate(ps)
    rc, *_ = flatten(lib, ps)
    assert rc == code
    assert lib.pt_last_error() != b""


def test_image_outside_atlas_rejected(lib):
    ps, _ = S.mixed_scene()
    for i in range(ps.n_textures):
        if ps.textures[i].kind == abi.PT_TEX_IMAGE:
            ps.textures[i].offset = 10 ** 7
    assert flatten(lib, ps)[0] == abi.PT_ERR_BAD_SCENE


def test_constructor_semantics():
    assert metal_material((1, 1, 1), 7.0).fuzz == 1.0 and metal_material((1, 1, 1), -1).fuzz == 0.0  # material.hpp:37
    s = sphere((1, 2, 3), 0.5, lambertian_material((1, 1, 1)))
    assert s.center0 == s.center1 and s.time0 == s.time1 == 0.0  # sphere.hpp:30-36
    cm = constant_medium(s, 4, (1, 1, 1))
    assert cm.neg_inv_density == -0.25
    with pytest.raises(TypeError):
        constant_medium(xy_rect(0, 1, 0, 1, 0, lambertian_material((1, 1, 1))), 1, (1, 1, 1))
    ps = pack([s, s, cm])
    assert ps.n_materials == 2 and ps.n_textures == 1  # value-equal materials/textures are shared


def test_scene_generators_are_deterministic():
    a, _ = scenes.build("smoke")
    b, _ = scenes.build("smoke")
    assert bytes(a.hittables) == bytes(b.hittables) and a.n_hittables > 480
    kinds = a.kinds()
    assert kinds[0] == abi.PT_HIT_SPHERE and kinds[-1] == abi.PT_HIT_CONSTANT_MEDIUM and kinds.count(abi.PT_HIT_TRIANGLE) == 4
    t, _ = scenes.triangle_mesh_scene(1000)
    assert t.n_hittables == 1002
    f = np.frombuffer(bytes(t.hittables), dtype=scenes.hittable_dtype)
    tri = f[1:-1]["f"]
    assert np.abs(tri[:, 3:6] - tri[:, 0:3]).max() <= 0.15 + 1e-6


def test_no_gpu_is_reported_not_crashed(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ps, _ = S.cornell_scene()
    h = C.c_void_p()
    rc = lib.pt_scene_create(C.byref(ps.desc), C.byref(h))
    assert rc in (abi.PT_ERR_NO_DEVICE, abi.PT_ERR_HIP) and not h.value


def test_scene_fixture_roundtrip(tmp_path, orc):
    """scene_io: the .npz fixture holds the C-ABI tables byte for byte; a reloaded scene renders identically."""
    from path_tracer_amd import scene_io
    for name in ("mixed", "cornell", "empty"):
        ps, cam = S.ALL[name]()
        f = tmp_path / f"{name}.npz"
        scene_io.save_scene(str(f), ps, cam)
        ps2, cam2 = scene_io.load_scene(str(f))
        assert (ps2.n_hittables, ps2.n_materials, ps2.n_textures) == (ps.n_hittables, ps.n_materials, ps.n_textures)
        assert bytes(ps2.hittables)[:ps.n_hittables * 64] == bytes(ps.hittables)[:ps.n_hittables * 64]
        assert bytes(ps2.materials)[:ps.n_materials * 32] == bytes(ps.materials)[:ps.n_materials * 32]
        assert bytes(ps2.textures)[:ps.n_textures * 48] == bytes(ps.textures)[:ps.n_textures * 48]
        assert bytes(ps2.atlas)[:ps.atlas_bytes] == bytes(ps.atlas)[:ps.atlas_bytes]
        assert {k: tuple(v) if isinstance(v, list) else v for k, v in cam2.items()} == \
               {k: tuple(v) if isinstance(v, (list, tuple)) else v for k, v in cam.items()}
        c = scenes.make_camera(cam, 16, 12)
        orc.set_math(True)
        a, b = orc.render(ps, c.c, 16, 12, 2), orc.render(ps2, c.c, 16, 12, 2)
        assert a.tobytes() == b.tobytes()


def _blob(lib, ps, tuning=None):
    n, r, fl = C.c_int32(), C.c_int32(), C.c_int32()
    t = C.byref(tuning) if tuning is not None else None
    abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), t, None, 0, C.byref(n), C.byref(r), None, 0, C.byref(fl)), "pt_debug_flatten_tuned")
    blob = np.zeros(n.value * 4, np.float32)
    abi.check(lib.pt_debug_flatten_tuned(C.byref(ps.desc), t, blob.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n), C.byref(r),
                                         None, 0, C.byref(fl)), "pt_debug_flatten_tuned")
    return blob.tobytes(), fl.value


def test_tuning_struct_and_environment_give_identical_blobs(lib, monkeypatch):
    """what pt_scene_create builds no longer depends on the caller's environment only — a C-ABI caller sets the
    grid / pool thresholds through PtTuning; the PT_* variables stay as the override channel of tools/ (a NULL PtTuning = defaults +
    environment).  Every knob: the struct and the variable flatten to the same bytes, and an explicit struct ignores the environment."""
    body = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    body = body[body.index("typedef struct PtTuning {") + len("typedef struct PtTuning {"):body.index("} PtTuning;")]
    dwords, names = 0, []
    for decl in filter(None, (d.strip() for d in body.split(";"))):  # "int32_t a, b[3]" -> every field is a 4-byte scalar or an array of them
        for item in decl.split(None, 1)[1].split(","):
            m = re.fullmatch(r"\s*(\w+)(?:\[(\d+)\])?\s*", item)
            names.append(m.group(1)); dwords += int(m.group(2) or 1)
    assert C.sizeof(abi.PtTuning) == 4 * dwords and names == [f[0] for f in abi.PtTuning._fields_]  # header and abi.py agree field for field
    t0 = abi.tuning()
    assert t0.struct_size == C.sizeof(abi.PtTuning)
    smoke, _ = scenes.build("smoke")
    tri, _ = scenes.triangle_mesh_scene(n_triangles=3000)
    mixed, _ = S.ALL["mixed"]()
    base = {id(s): _blob(lib, s) for s in (smoke, tri, mixed)}
    assert _blob(lib, smoke, t0) == base[id(smoke)] and _blob(lib, tri, t0) == base[id(tri)]
    cases = [  # (environment, the same as PtTuning fields, scene)
        ({"PT_NO_GRID": "1"}, dict(sphere_grid=-1), smoke),
        ({"PT_GRID_M": "0.75", "PT_GRID_CELL": "3.43"}, dict(grid_margin=0.75, grid_cell=3.43), smoke),
        ({"PT_TRICULL": "1"}, dict(tri_min_run=256), tri),
        ({"PT_TRI_MIN": "1000", "PT_TRI_M": "20", "PT_TRI_MG": "48", "PT_TRI_CELL": "1.0", "PT_TRI_RES": "64,32,16"},
         dict(tri_min_run=1000, tri_M=20.0, tri_binned=-1, tri_cell=1.0, tri_res=(64, 32, 16)), tri),
        ({"PT_TRICULL": "1", "PT_NO_TRICULL": "1"}, dict(tri_min_run=256, tri_pool=-1), tri),
        ({"PT_POOL_ALWAYS": "1"}, dict(slab_pools=1), mixed),
        ({"PT_NO_BOXCULL": "1"}, dict(slab_pools=-1), scenes.build("cornell")[0]),
    ]
    for env, fields, ps in cases:
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            from_env = _blob(lib, ps)
            te = abi.PtTuning()
            lib.pt_tuning_from_env(C.byref(te))
            assert _blob(lib, ps, te) == from_env, env
            assert _blob(lib, ps, t0) == _blob(lib, ps, abi.tuning()), env  # an explicit struct: the environment is not consulted
        from_struct = _blob(lib, ps, abi.tuning(**fields))
        assert from_struct == from_env, (env, fields)
        if ps in (smoke, tri):
            assert from_struct != base[id(ps)] or "PT_NO_TRICULL" in env, (env, "the knob changed nothing")
    # an unknown struct_size is an error code, a shorter (older) struct is accepted
    bad = abi.tuning(); bad.struct_size = 4
    n = C.c_int32()
    assert lib.pt_debug_flatten_tuned(C.byref(smoke.desc), C.byref(bad), None, 0, C.byref(n), None, None, 0, None) == abi.PT_ERR_INVALID_ARG
    old = abi.tuning(sphere_grid=-1); old.struct_size = 16
    with monkeypatch.context() as m:
        m.setenv("PT_NO_GRID", "1")
        assert _blob(lib, smoke, old) == _blob(lib, smoke)
