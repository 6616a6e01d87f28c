"""ctypes mirror of include/pt_render.h — the C-ABI drop-in boundary of render().

The structs are declared field-for-field as in the header; `load_library()` opens the
in-tree `libpt_render.so` built by `__graft_entry__.build()` (hipcc, gfx950) and fails
loudly when it is missing: there is no CPU fallback in the product path.

Reference boundary being replaced: `render<W,H,S>(queue, frame_buf, hittables, cam)`
(/root/reference include/render.hpp:141-160).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PT_ABI_VERSION = 2  # (1: rounds 1-5; the table structs did not change: scene_io loads both)

# tags — same numbering as the reference's std::variant alternatives
PT_HIT_SPHERE, PT_HIT_XY_RECT, PT_HIT_TRIANGLE, PT_HIT_BOX, PT_HIT_CONSTANT_MEDIUM = 0, 1, 2, 3, 4
PT_HIT_XZ_RECT, PT_HIT_YZ_RECT = 5, 6  # extension (rectangle.hpp:54,92 exist but are not hittable_t members)
PT_HIT_KIND_COUNT = 7
PT_MAT_LAMBERTIAN, PT_MAT_METAL, PT_MAT_DIELECTRIC, PT_MAT_LIGHTSOURCE, PT_MAT_ISOTROPIC = 0, 1, 2, 3, 4
PT_TEX_CHECKER, PT_TEX_SOLID, PT_TEX_IMAGE = 0, 1, 2

PT_TILE = 8
PT_TILE_PIXELS = 64
PT_TRI_MOLLER_TRUMBORE, PT_TRI_BADOUEL = 0, 1  # PtHittable.strategy of a triangle (triangle.hpp:102-103)
PT_FLAG_NONE = 0
PT_FLAG_NO_LDS = 1
PT_FLAG_FORCE_STREAM = 2
PT_FLAG_NO_FASTDIV = 4
PT_FLAG_TILE_GRANULAR = 8
PT_FLAG_PIXEL_GRANULAR = 16
PT_FLAG_NO_LPT = 32
PT_FLAG_NO_COOP = 64
PT_FLAG_FORCE_COOP = 128
PT_FLAG_NO_SPLIT = 256
PT_FLAG_FAST_RNG = 512  # opt-in decorrelated RNG streams: NOT the reference's image (include/pt_render.h)
PT_FAST_CHUNK_SPP = 64
PT_FLAG_SINGLE_STREAM = 1024  # the reference's USE_SINGLE_TASK executor: one RNG stream for the whole frame (small frames)

PT_OK, PT_ERR_INVALID_ARG, PT_ERR_BAD_SCENE, PT_ERR_HIP, PT_ERR_NO_DEVICE, PT_ERR_TOO_LARGE = range(6)
PT_BOUNCE_MISS, PT_BOUNCE_SCATTERED, PT_BOUNCE_ABSORBED = 0, 1, 2


class PtHittable(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("boundary_kind", C.c_int32),
                ("strategy", C.c_int32), ("f", C.c_float * 12)]


class PtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("color", C.c_float * 3),
                ("param", C.c_float), ("reserved", C.c_int32 * 2)]


class PtTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("color0", C.c_float * 3), ("color1", C.c_float * 3),
                ("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint32),
                ("freq", C.c_float), ("reserved", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [("hittables", C.POINTER(PtHittable)), ("n_hittables", C.c_int32),
                ("materials", C.POINTER(PtMaterial)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(PtTexture)), ("n_textures", C.c_int32),
                ("reserved", C.c_int32),
                ("atlas", C.POINTER(C.c_uint8)), ("atlas_bytes", C.c_uint64)]


class PtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3),
                ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class PtRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("depth", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_uint32),
                ("reserved", C.c_int32)]


class PtTuning(C.Structure):
    """include/pt_render.h PtTuning: performance-only knobs (same image under all of them).  Zero = the library's default."""
    _fields_ = [("struct_size", C.c_int32), ("sphere_grid", C.c_int32), ("grid_margin", C.c_float), ("grid_cell", C.c_float),
                ("slab_pools", C.c_int32), ("tri_pool", C.c_int32), ("tri_min_run", C.c_int32),
                ("tri_M", C.c_float), ("tri_binned", C.c_int32), ("tri_cell", C.c_float), ("tri_res", C.c_int32 * 3),
                ("generic_materials", C.c_int32), ("blocks_per_cu", C.c_int32), ("cold_state", C.c_int32),
                ("wide_log2_group", C.c_int32), ("split_tiles_mode", C.c_int32), ("split_tiles", C.c_int32),
                ("lpt_by_max", C.c_int32), ("probe_spp_max", C.c_int32), ("grid_min_tiles", C.c_int32),
                ("model_fixed", C.c_float), ("model_chain", C.c_float), ("scatter_log", C.c_int32), ("scatter_mode", C.c_int32),
                ("lanes_cap", C.c_int32), ("grid_walk", C.c_int32), ("heavy_tiles", C.c_int32),
                ("tri_rho", C.c_float * 2), ("tri_budget_mb", C.c_int32), ("tri_rho2", C.c_float), ("probe_resume", C.c_int32), ("chain_priority", C.c_int32), ("tri_cache", C.c_int32), ("sphere_merge", C.c_int32)]


def tuning(**fields) -> "PtTuning":
    """A PtTuning with the library's defaults (pt_tuning_init) and the given fields set."""
    t = PtTuning()
    load_library().pt_tuning_init(C.byref(t))
    for k, v in fields.items():
        if k == "tri_res":
            t.tri_res[:] = list(v)
        elif k == "tri_rho":
            t.tri_rho[:] = list(v)
        else:
            setattr(t, k, v)
    return t


class PtBounceIn(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("time", C.c_float),
                ("rng_state", C.c_uint32), ("attenuation", C.c_float * 3)]


class PtBounceOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("hittable", C.c_int32), ("material", C.c_int32), ("front_face", C.c_int32),
                ("t", C.c_float), ("p", C.c_float * 3), ("normal", C.c_float * 3), ("u", C.c_float), ("v", C.c_float),
                ("color", C.c_float * 3), ("sc_origin", C.c_float * 3), ("sc_dir", C.c_float * 3),
                ("sc_time", C.c_float), ("rng_state", C.c_uint32)]


class PtCameraRay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("time", C.c_float), ("rng_state", C.c_uint32)]


assert C.sizeof(PtHittable) == 64 and C.sizeof(PtMaterial) == 32 and C.sizeof(PtTexture) == 48
assert C.sizeof(PtCamera) == 96 and C.sizeof(PtRenderParams) == 32

_FP = C.POINTER(C.c_float)
_SCENE_P = C.c_void_p

# name -> (restype, argtypes): every entry point include/pt_render.h declares
SIGNATURES = {
    "pt_abi_version": (C.c_int, []),
    "pt_error_string": (C.c_char_p, [C.c_int]),
    "pt_last_error": (C.c_char_p, []),
    "pt_camera_init": (C.c_int, [C.POINTER(PtCamera), _FP, _FP, _FP, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float]),
    "pt_scene_create": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(_SCENE_P)]),
    "pt_scene_create_tuned": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(PtTuning), C.POINTER(_SCENE_P)]),
    "pt_tuning_init": (None, [C.POINTER(PtTuning)]),
    "pt_tuning_from_env": (None, [C.POINTER(PtTuning)]),
    "pt_debug_flatten_tuned": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(PtTuning), _FP, C.c_int64, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), _FP, C.c_int64, C.POINTER(C.c_int32)]),
    "pt_scene_destroy": (None, [_SCENE_P]),
    "pt_scene_device_bytes": (C.c_int64, [_SCENE_P]),
    "pt_build_id": (C.c_char_p, []),
    "pt_scene_reserve": (C.c_int, [_SCENE_P, C.POINTER(PtRenderParams)]),
    "pt_fast_seed": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "pt_framebuffer_floats": (C.c_int64, [C.POINTER(PtRenderParams)]),
    "pt_shard_tiles": (C.c_int32, [C.POINTER(PtRenderParams)]),
    "pt_render": (C.c_int, [_SCENE_P, C.POINTER(PtCamera), C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_render_timed": (C.c_int, [_SCENE_P, C.POINTER(PtCamera), C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_float)]),
    "pt_render_host": (C.c_int, [_SCENE_P, C.POINTER(PtCamera), C.POINTER(PtRenderParams), _FP]),
    "pt_unshard_tiles": (C.c_int, [C.c_void_p, C.POINTER(PtRenderParams), C.c_void_p, C.c_void_p]),
    "pt_tonemap_rgb8": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pt_debug_bounce": (C.c_int, [_SCENE_P, C.POINTER(PtBounceIn), C.POINTER(PtBounceOut), C.c_int32]),
    "pt_debug_camera_rays": (C.c_int, [C.POINTER(PtCamera), C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_uint32), C.POINTER(PtCameraRay), C.c_int32]),
    "pt_debug_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pt_debug_last_launch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pt_debug_flatten": (C.c_int, [C.POINTER(PtSceneDesc), _FP, C.c_int64, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), _FP, C.c_int64, C.POINTER(C.c_int32)]),
    "pt_debug_math": (C.c_int, [C.c_int32, _FP, _FP, _FP, C.c_int64]),
    "pt_debug_sphere_texel": (C.c_int, [_FP, C.c_int64, C.c_float, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint8), _FP]),
    "pt_debug_tri_pool": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(C.c_int32)]),
    "pt_debug_flatten_pool": (C.c_int, [C.POINTER(PtSceneDesc), C.POINTER(PtTuning), C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
}

LIB_NAME = "libpt_render.so"
_lib = None


class PtError(RuntimeError):
    """A pt_* entry point returned a non-zero code (the reference's `void` + asserts, made explicit)."""

    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: error {code}" + (f" ({detail})" if detail else ""))


def library_path() -> Path:
    override = os.environ.get("PT_RENDER_LIB")
    return Path(override) if override else Path(__file__).resolve().parent / LIB_NAME


def override_is_older_build(path, lib, name) -> bool:
    """A/B tooling only (tools/abn.sh loads an OLDER build of the extension through PT_RENDER_LIB): entry points added since
    that build may be missing there.  The in-tree library must export everything include/pt_render.h declares."""
    return bool(os.environ.get("PT_RENDER_LIB")) and os.environ.get("PT_RENDER_LIB_ALLOW_OLDER") == "1" and not hasattr(lib, name)


def load_library() -> C.CDLL:
    """Open the HIP extension.  Fails loudly: the product has no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise ImportError(
            f"{path} is missing: the gfx950 HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc cross-compiles without a GPU).")
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Two
    # HIP runtimes in one process cannot both see the GPU, and torch asks for its copy by file name, so the
    # order matters: with torch loaded first the loader binds this library's libamdhip64.so.7 dependency to
    # torch's copy by SONAME; loaded the other way round the process would end up with two runtimes.
    try:
        import torch  # noqa: F401  (plumbing: device memory, streams, RCCL)
    except ImportError:
        pass  # torch-free hosts bind to /opt/rocm/lib through the library's RUNPATH
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        if override_is_older_build(path, lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    older_ok = bool(os.environ.get("PT_RENDER_LIB")) and os.environ.get("PT_RENDER_LIB_ALLOW_OLDER") == "1"  # (A/B tooling: an older build beside this tree)
    if lib.pt_abi_version() != PT_ABI_VERSION and not (older_ok and lib.pt_abi_version() == 1):
        raise ImportError(f"{path}: ABI version {lib.pt_abi_version()} != {PT_ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, where: str) -> None:
    if code != PT_OK:
        lib = load_library()
        detail = lib.pt_last_error().decode() or lib.pt_error_string(code).decode()
        raise PtError(code, where, detail)
