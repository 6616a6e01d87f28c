"""ctypes binding of oracle/liboracle.so (oracle/pt_oracle.h).  TEST INFRASTRUCTURE ONLY.

Reuses the C-ABI struct declarations of path_tracer_amd.abi: oracle and product consume the
same PtSceneDesc / PtCamera / PtRenderParams tables, so one scene dump feeds both sides."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from path_tracer_amd import abi

HERE = Path(__file__).resolve().parent
LIB = HERE / "liboracle.so"
REF_KAT = HERE / "_ref" / "xorshift_kat"
REF_VISIT_KAT = REF_KAT.parent / "visit_kat"  # the reference's own visit.hpp (dev_visit), compiled as it lies (oracle/Makefile)


class OrcCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("rng_draws", C.c_uint64),
                ("tests", C.c_uint64 * abi.PT_HIT_KIND_COUNT), ("accepts", C.c_uint64 * abi.PT_HIT_KIND_COUNT),
                ("rect_tests", C.c_uint64), ("sphere_tests", C.c_uint64), ("scatters", C.c_uint64 * 5),
                ("end_sky", C.c_uint64), ("end_emit", C.c_uint64), ("end_depth", C.c_uint64),
                ("rect_exit", C.c_uint64 * 3), ("tri_exit", C.c_uint64 * 5), ("sphere_exit", C.c_uint64 * 3),
                ("sphere_moving", C.c_uint64), ("tex_evals", C.c_uint64 * 3)]

    def as_dict(self) -> dict:
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


_FP = C.POINTER(C.c_float)
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", str(HERE), "-s"], check=True)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        build()
    lib = C.CDLL(str(LIB))
    lib.orc_set_math.argtypes = [C.c_int]
    lib.orc_get_math.restype = C.c_int
    lib.orc_xorshift32.restype = C.c_uint32
    lib.orc_xorshift32.argtypes = [C.POINTER(C.c_uint32)]
    lib.orc_float_t.restype = C.c_float
    lib.orc_float_t.argtypes = [C.POINTER(C.c_uint32)]
    for n in ("orc_unit_vec", "orc_in_unit_ball", "orc_in_unit_disk"):
        getattr(lib, n).argtypes = [C.POINTER(C.c_uint32), _FP]
        getattr(lib, n).restype = None
    lib.orc_camera_init.argtypes = [C.POINTER(abi.PtCamera), _FP, _FP, _FP] + [C.c_float] * 6
    lib.orc_camera_init.restype = None
    lib.orc_render.argtypes = [C.POINTER(abi.PtSceneDesc), C.POINTER(abi.PtCamera), C.POINTER(abi.PtRenderParams),
                               _FP, C.POINTER(OrcCounters)]
    lib.orc_render_rows.argtypes = [C.POINTER(abi.PtSceneDesc), C.POINTER(abi.PtCamera),
                                    C.POINTER(abi.PtRenderParams), C.c_int32, C.c_int32, _FP, C.POINTER(OrcCounters)]
    lib.orc_render_pixels.argtypes = [C.POINTER(abi.PtSceneDesc), C.POINTER(abi.PtCamera),
                                      C.POINTER(abi.PtRenderParams), C.POINTER(C.c_int32), C.c_int32, _FP]
    lib.orc_bounce.argtypes = [C.POINTER(abi.PtSceneDesc), C.POINTER(abi.PtBounceIn), C.POINTER(abi.PtBounceOut),
                               C.c_int32, C.c_int32]
    lib.orc_camera_rays.argtypes = [C.POINTER(abi.PtCamera), C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_uint32), C.POINTER(abi.PtCameraRay), C.c_int32]
    lib.orc_math.argtypes = [C.c_int32, _FP, _FP, _FP, C.c_int64]
    lib.orc_tonemap_rgb8.argtypes = [_FP, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
    lib.orc_tonemap_rgb8.restype = None
    lib.orc_max_threads.restype = C.c_int
    lib.orc_set_threads.restype = None
    lib.orc_set_threads.argtypes = [C.c_int]
    _lib = lib
    return lib


def set_math(portable: bool) -> None:
    load().orc_set_math(1 if portable else 0)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def params(width, height, samples, depth=50, shard_index=0, shard_count=1, flags=0) -> abi.PtRenderParams:
    return abi.PtRenderParams(width, height, samples, depth, shard_index, shard_count, flags, 0)


def camera_init(look_from, look_at, vup, vfov, aspect, aperture, focus_dist, t0=0.0, t1=0.0) -> abi.PtCamera:
    cam = abi.PtCamera()
    a3 = lambda v: (C.c_float * 3)(*[float(np.float32(x)) for x in v])  # noqa: E731
    load().orc_camera_init(C.byref(cam), a3(look_from), a3(look_at), a3(vup), np.float32(vfov), np.float32(aspect),
                           np.float32(aperture), np.float32(focus_dist), np.float32(t0), np.float32(t1))
    return cam


def render(packed, cam: abi.PtCamera, width, height, samples, depth=50, shard_index=0, shard_count=1,
           counters: bool = False, flags: int = 0):
    """flags: only PT_FLAG_FAST_RNG means anything to the oracle (the opt-in decorrelated mode's own checker)."""
    lib = load()
    p = params(width, height, samples, depth, shard_index, shard_count, flags)
    if shard_count == 1:
        fb = np.zeros((height, width, 3), dtype=np.float32)
    else:
        tiles = ((width + 7) // 8) * ((height + 7) // 8)
        fb = np.zeros(((tiles + shard_count - 1) // shard_count, 64, 3), dtype=np.float32)
    ctr = OrcCounters()
    rc = lib.orc_render(C.byref(packed.desc), C.byref(cam), C.byref(p), _fp(fb), C.byref(ctr) if counters else None)
    if rc:
        raise RuntimeError(f"orc_render: error {rc}")
    return (fb, ctr) if counters else fb


def render_rows(packed, cam: abi.PtCamera, width, height, samples, y0, y1, depth=50, counters: bool = False):
    lib = load()
    p = params(width, height, samples, depth)
    fb = np.zeros((y1 - y0, width, 3), dtype=np.float32)
    ctr = OrcCounters()
    rc = lib.orc_render_rows(C.byref(packed.desc), C.byref(cam), C.byref(p), y0, y1, _fp(fb),
                             C.byref(ctr) if counters else None)
    if rc:
        raise RuntimeError(f"orc_render_rows: error {rc}")
    return (fb, ctr) if counters else fb


def render_pixels(packed, cam: abi.PtCamera, width, height, samples, xy: np.ndarray, depth=50, flags: int = 0) -> np.ndarray:
    lib = load()
    p = params(width, height, samples, depth, flags=flags)
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = np.zeros((len(xy), 3), dtype=np.float32)
    rc = lib.orc_render_pixels(C.byref(packed.desc), C.byref(cam), C.byref(p), xy.ctypes.data_as(C.POINTER(C.c_int32)),
                               len(xy), _fp(out))
    if rc:
        raise RuntimeError(f"orc_render_pixels: error {rc}")
    return out


def render_pixels_rays(packed, cam: abi.PtCamera, width, height, samples, xy: np.ndarray, depth=50):
    """(colours [n][3], rays traced per pixel [n]) — the per-pixel chain lengths behind DESIGN.md §6."""
    lib = load()
    p = params(width, height, samples, depth)
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = np.zeros((len(xy), 3), dtype=np.float32)
    rays = np.zeros(len(xy), dtype=np.uint64)
    lib.orc_render_pixels_rays.restype = C.c_int
    rc = lib.orc_render_pixels_rays(C.byref(packed.desc), C.byref(cam), C.byref(p), xy.ctypes.data_as(C.POINTER(C.c_int32)),
                                    len(xy), _fp(out), rays.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc:
        raise RuntimeError(f"orc_render_pixels_rays: error {rc}")
    return out, rays


def bounce(packed, recs_in):
    lib = load()
    n = len(recs_in)
    out = (abi.PtBounceOut * max(1, n))()
    rc = lib.orc_bounce(C.byref(packed.desc), recs_in, out, n, 0)
    if rc:
        raise RuntimeError(f"orc_bounce: error {rc}")
    return out


def camera_rays(cam: abi.PtCamera, width, height, xy: np.ndarray, rng_in: np.ndarray):
    lib = load()
    n = len(rng_in)
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    rng_in = np.ascontiguousarray(rng_in, dtype=np.uint32)
    out = (abi.PtCameraRay * max(1, n))()
    lib.orc_camera_rays(C.byref(cam), width, height, xy.ctypes.data_as(C.POINTER(C.c_int32)),
                        rng_in.ctypes.data_as(C.POINTER(C.c_uint32)), out, n)
    return out


def math(op: int, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    bb = np.ascontiguousarray(b, dtype=np.float32) if b is not None else np.zeros_like(a)
    out = np.empty_like(a)
    rc = lib.orc_math(op, _fp(a), _fp(bb), _fp(out), a.size)
    if rc:
        raise RuntimeError(f"orc_math: error {rc}")
    return out


def tonemap_rgb8(fb: np.ndarray) -> np.ndarray:
    lib = load()
    h, w, _ = fb.shape
    fb = np.ascontiguousarray(fb, dtype=np.float32)
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib.orc_tonemap_rgb8(_fp(fb), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def xorshift_stream(seed: int, count: int) -> list:
    lib = load()
    s = C.c_uint32(seed)
    return [lib.orc_xorshift32(C.byref(s)) for _ in range(count)]
