"""The product's HOST-side builders under AddressSanitizer + UBSan: pt_flatten.hpp + pt_tripool.hpp build the
sphere grids, slab pools, the triangle pools' fine grids (neighbour bits), direction maps, Morton-ordered copies and quantised records with raw offsets, and only the ORACLE had a
sanitizer build.  A host-only TU (tests/cpp/flatten_host.cpp) is compiled with g++ -fsanitize=address,undefined and flattens the
Cornell-style scene, the 496-hittable scene, the 100 k-triangle mesh (with its triangle pool) and 50 fuzz scenes — degenerate and
duplicated triangles, zero-radius spheres, single-element and empty runs, boxes of no volume, non-finite coordinates — in a subprocess
with libasan preloaded; every blob must equal the shipped library's own (pt_debug_flatten), byte for byte.
GPU sanitizers are not available on this pool; the reference's analogue is CMakeLists.txt:76-90."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from path_tracer_amd import abi, scenes
from path_tracer_amd.scene import hittable_dtype, pack_tables
san = C.CDLL(sys.argv[2])
san.flat_check.argtypes = [C.POINTER(abi.PtSceneDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_int64]
lib = abi.load_library()  # the shipped (uninstrumented) flattener, host-only entry point


def product_blob(ps):
    n, r, fl = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.pt_debug_flatten(C.byref(ps.desc), None, 0, C.byref(n), C.byref(r), None, 0, C.byref(fl))
    if rc:
        return rc, None
    blob = np.zeros(n.value * 4, np.float32)
    abi.check(lib.pt_debug_flatten(C.byref(ps.desc), blob.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n), C.byref(r), None, 0, C.byref(fl)), "flatten")
    return 0, blob


def check(name, ps, tri_min=0, compare=True, compare_pool=True):
    n, h, st, npool = C.c_int64(), C.c_uint64(), (C.c_int32 * 4)(), C.c_int64()
    rc = san.flat_check(C.byref(ps.desc), 1, 1, 1, tri_min, C.byref(n), C.byref(h), st, None, 0, C.byref(npool), None, 0)
    prc, ref = product_blob(ps) if compare else (rc, None)
    assert rc == prc, (name, rc, prc)
    if rc == 0 and compare:
        blob = np.zeros(n.value * 4, np.float32)
        pool = np.zeros(max(npool.value, 1) * 4, np.float32) if compare_pool else None
        FP = C.POINTER(C.c_float)
        assert san.flat_check(C.byref(ps.desc), 1, 1, 1, tri_min, C.byref(n), C.byref(h), st, blob.ctypes.data_as(FP), n.value, C.byref(npool),
                              pool.ctypes.data_as(FP) if compare_pool else None, npool.value if compare_pool else 0) == 0
        assert blob.tobytes() == ref.tobytes(), name
        # the triangle pools' tables (their own buffer): the shipped library's, byte for byte
        pn = C.c_int64()
        abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), None, None, 0, C.byref(pn)), "pool size")
        assert pn.value == npool.value, (name, pn.value, npool.value)
        if compare_pool and pn.value:
            pref = np.zeros(pn.value * 4, np.float32)
            abi.check(lib.pt_debug_flatten_pool(C.byref(ps.desc), None, pref.ctypes.data_as(FP), pn.value, C.byref(pn)), "pool")
            assert pool.tobytes() == pref.tobytes(), name + " (pool tables)"
    return rc, list(st)


for name in ("cornell", "smoke"):
    ps, _ = scenes.build(name)
    rc, st = check(name, ps)
    assert rc == 0 and st[0] >= 3, (name, st)
ps, _ = scenes.build("triangles", n_triangles=100_000)
rc, st = check("triangles 100k", ps, compare_pool=False)  # (its maps are ~2 GB: built and checksummed under the sanitizers, sizes compared)
assert rc == 0 and st[3] == 100_000, st  # the whole run sits in a triangle pool
ps, _ = scenes.build("triangles", n_triangles=6_000)
rc, st = check("triangles 6k", ps)                        # the same builders at a size whose tables are compared byte for byte
assert rc == 0 and st[3] == 6_000, st

# fuzz: what a caller can put into the tables, including what no sane scene contains
os.environ["PT_TRICULL"] = "1"  # the shipped library's threshold for the comparison blob (pt_debug_flatten reads the knob): pools from 256 triangles
mat = abi.PtMaterial(); mat.kind = abi.PT_MAT_LAMBERTIAN; mat.texture = 0
tex = abi.PtTexture(); tex.kind = abi.PT_TEX_SOLID; tex.color0[:] = [0.5, 0.5, 0.5]
for seed in range(50):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([0, 1, 2, 3, 47, 48, 49, 300, 1500]))
    h = np.zeros(n, dtype=hittable_dtype)
    kinds = rng.choice([abi.PT_HIT_SPHERE, abi.PT_HIT_XY_RECT, abi.PT_HIT_TRIANGLE, abi.PT_HIT_BOX, abi.PT_HIT_CONSTANT_MEDIUM, abi.PT_HIT_XZ_RECT,
                        abi.PT_HIT_YZ_RECT], size=n, p=[0.3, 0.05, 0.4, 0.1, 0.05, 0.05, 0.05])
    if seed % 3 == 0:   # long runs of one kind (grids / pools / slab pools get built)
        kinds = np.sort(kinds)
    if seed % 7 == 1 and n:
        kinds[:] = abi.PT_HIT_TRIANGLE
    if seed % 7 == 2 and n:
        kinds[:] = abi.PT_HIT_SPHERE
    h["kind"] = kinds
    h["boundary_kind"] = rng.choice([abi.PT_HIT_SPHERE, abi.PT_HIT_BOX], size=n)
    scale = float(rng.choice([1e-2, 1.0, 1e2]))
    f = (rng.random((n, 12)).astype(np.float32) - 0.5) * scale
    for i in range(n):
        k, r = kinds[i], rng.random()
        if k == abi.PT_HIT_SPHERE:
            f[i, 3:6] = f[i, 0:3] if r < 0.6 else f[i, 3:6]      # centre1
            f[i, 6] = 0.0 if r < 0.1 else abs(f[i, 6]) * 0.1      # zero-radius spheres
            f[i, 7:9] = (0.0, 0.0) if r < 0.6 else (0.0, 1.0)     # shutter
        elif k == abi.PT_HIT_TRIANGLE:
            if r < 0.1: f[i, 3:6] = f[i, 0:3]                     # degenerate: two vertices coincide
            elif r < 0.15: f[i, 3:9] = np.tile(f[i, 0:3], 2)      # a point
            elif r < 0.25: f[i, 6:9] = f[i, 0:3] + 2 * (f[i, 3:6] - f[i, 0:3])  # collinear
            elif r < 0.35 and i: f[i] = f[i - 1]                  # duplicate of the previous one
            else: f[i, 3:9] = np.tile(f[i, 0:3], 2) + (rng.random(6).astype(np.float32) - 0.5) * 0.05 * scale
        elif k in (abi.PT_HIT_BOX, abi.PT_HIT_CONSTANT_MEDIUM):
            lo, hi = np.minimum(f[i, 0:3], f[i, 3:6]), np.maximum(f[i, 0:3], f[i, 3:6])
            if r < 0.15: hi = lo.copy()                           # a box of no volume
            f[i, 0:3], f[i, 3:6] = lo, hi
            f[i, 6] = abs(f[i, 6]) + 0.01; f[i, 9] = -1.0 / 0.5
        else:
            a = np.sort(f[i, 0:2]); b = np.sort(f[i, 2:4]); f[i, 0:2] = a; f[i, 2:4] = b
    if seed % 11 == 5 and n:                                      # non-finite coordinates here and there
        idx = rng.integers(0, n, max(1, n // 20))
        f[idx, rng.integers(0, 9, len(idx))] = rng.choice([np.inf, -np.inf, np.nan], len(idx))
    h["f"] = f
    ps = pack_tables(h, [mat], [tex])
    rc, st = check(f"fuzz {seed}", ps, tri_min=256)
    assert rc == 0, (seed, rc)
print("FLATTEN_SANITIZED_OK")
'''


def test_flatten_and_tripool_under_asan_ubsan(tmp_path):
    so = tmp_path / "libflatten_asan.so"
    cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
           "-shared", "-o", str(so), str(ROOT / "tests" / "cpp" / "flatten_host.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for k in list(env):
        if k.startswith("PT_"):
            del env[k]
    out = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), str(so)], capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0 and "FLATTEN_SANITIZED_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-4000:]
