"""Frozen function-level known-answer tables (tests/golden/function_tables.npz, made by tests/golden/make_function_tables.py
from the oracle in portable-math mode): per primitive x material, ONE hit() + emitted()/scatter() per row; per camera, the
constructor's fields and get_ray rows.  CPU: the oracle still produces them (drift guard for the checker).  GPU: the device
produces them (pt_debug_bounce / pt_debug_camera_rays), bit for bit."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import function_tables as FT
from conftest import assert_bit_identical
from path_tracer_amd import abi, scenes

TABLES = Path(__file__).resolve().parent / "golden" / "function_tables.npz"
FLOAT_FIELDS = {"t", "p", "normal", "u", "v", "color", "sc_origin", "sc_dir", "sc_time"}


@pytest.fixture(scope="module")
def tables():
    with np.load(TABLES) as z:
        return {k: z[k].copy() for k in z.files}


def as_records(raw: np.ndarray, ctype):
    n = raw.size // C.sizeof(ctype)
    return (ctype * n).from_buffer_copy(raw.tobytes())


def compare(got, ref, what, with_uv):
    n = len(ref)
    for name, _ in abi.PtBounceOut._fields_:
        if name in ("u", "v") and not with_uv:
            continue  # the device tracks u,v only where an image texture can read them (include/pt_render.h)
        g = np.array([np.ctypeslib.as_array(getattr(got[k], name)) if hasattr(getattr(got[k], name), "__len__") else getattr(got[k], name) for k in range(n)])
        r = np.array([np.ctypeslib.as_array(getattr(ref[k], name)) if hasattr(getattr(ref[k], name), "__len__") else getattr(ref[k], name) for k in range(n)])
        if name in FLOAT_FIELDS:
            assert_bit_identical(g.astype(np.float32), r.astype(np.float32), f"{what}.{name}")
        else:
            assert (g == r).all(), f"{what}.{name}: rows {np.argwhere(g != r)[:5].tolist()} differ"


CASES = list(FT.cases())


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_frozen_hit_and_scatter_tables(orc, tables, name):
    ps, region = FT.cases()[name]
    recs = FT.rays(name, region)
    assert bytes(recs) == tables[f"bounce_in/{name}"].tobytes(), "the recorded inputs are regenerated deterministically"
    orc.set_math(True)
    got = orc.bounce(ps, recs)
    ref = as_records(tables[f"bounce_out/{name}"], abi.PtBounceOut)
    assert len(ref) == FT.N_RAYS
    compare(got, ref, name, with_uv=True)
    st = [ref[k].status for k in range(len(ref))]
    assert abi.PT_BOUNCE_MISS in st and any(s != abi.PT_BOUNCE_MISS for s in st)  # each table has hits and misses


@pytest.mark.parametrize("name", list(FT.CAMERAS))
def test_oracle_reproduces_the_frozen_camera_tables(orc, tables, name):
    look_from, look_at, vup, vfov, aperture, focus, t0, t1, w, h = FT.CAMERAS[name]
    if focus is None:
        d = np.float32(look_at) - np.float32(look_from)
        focus = float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
    args = dict(look_from=look_from, look_at=look_at, vup=vup, vfov=vfov, aperture=aperture, focus_dist=focus, time0=t0, time1=t1)
    cam = scenes.make_camera(args, w, h)  # pt_camera_init (the product's host code, no GPU needed)
    assert bytes(cam.c) == tables[f"camera_fields/{name}"].tobytes()
    a3 = lambda v: [float(np.float32(x)) for x in v]  # noqa: E731
    ocam = orc.camera_init(a3(look_from), a3(look_at), a3(vup), vfov, float(np.float32(w) / np.float32(h)), aperture, focus, t0, t1)
    assert bytes(ocam) == tables[f"camera_fields/{name}"].tobytes()
    xy, st = FT.camera_inputs(name)
    assert bytes(orc.camera_rays(cam.c, w, h, xy, st)) == tables[f"camera_rays/{name}"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_reproduces_the_frozen_hit_and_scatter_tables(lib, tables, name):
    from path_tracer_amd import render as R
    ps, region = FT.cases()[name]
    recs = as_records(tables[f"bounce_in/{name}"], abi.PtBounceIn)
    out = (abi.PtBounceOut * len(recs))()
    ds = R.DeviceScene(ps)
    abi.check(lib.pt_debug_bounce(ds.handle, recs, out, len(recs)), "pt_debug_bounce")
    has_image = any(ps.textures[i].kind == abi.PT_TEX_IMAGE for i in range(ps.n_textures))
    compare(out, as_records(tables[f"bounce_out/{name}"], abi.PtBounceOut), name, with_uv=has_image and "image" in name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FT.CAMERAS))
def test_device_reproduces_the_frozen_camera_tables(lib, tables, name):
    w, h = FT.CAMERAS[name][8:10]
    cam = abi.PtCamera.from_buffer_copy(tables[f"camera_fields/{name}"].tobytes())
    xy, st = FT.camera_inputs(name)
    out = (abi.PtCameraRay * len(st))()
    abi.check(lib.pt_debug_camera_rays(C.byref(cam), w, h, xy.ctypes.data_as(C.POINTER(C.c_int32)),
                                       st.ctypes.data_as(C.POINTER(C.c_uint32)), out, len(st)), "pt_debug_camera_rays")
    assert bytes(out) == tables[f"camera_rays/{name}"].tobytes()
