"""Rays as the renderer meets them: paths followed through a scene with the CPU oracle (orc.bounce: every scattered ray is
an input of the next generation — origins ON surfaces, inside glass, grazing directions) and the device's hit_world + shade
(pt_debug_bounce) checked on every ray of every generation.  Far more rays per second than a framebuffer comparison, and a
mismatch names the ray.  Used by tests/test_gpu_fuzz.py and tools/bounce_hunt.py."""
import numpy as np

from path_tracer_amd import abi
from path_tracer_amd import render as R


def follow_paths(lib, orc, ps, cam_c, w, h, n, generations, seed, verbose=False):
    """Returns (rays checked, list of mismatch descriptions)."""
    ds = R.DeviceScene(ps)
    orc.set_math(True)
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], 1).astype(np.int32)
    st = rng.integers(1, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    rays = orc.camera_rays(cam_c, w, h, xy, st)
    recs = (abi.PtBounceIn * n)()
    for k in range(n):
        recs[k].origin[:] = list(rays[k].origin); recs[k].dir[:] = list(rays[k].dir); recs[k].time = rays[k].time
        recs[k].rng_state = rays[k].rng_state; recs[k].attenuation[:] = [1.0, 1.0, 1.0]
    bad, checked = [], 0
    f32 = lambda x: np.float32(x).tobytes()
    for g in range(generations):
        m = len(recs)
        if m == 0:
            break
        out = (abi.PtBounceOut * m)()
        abi.check(lib.pt_debug_bounce(ds.handle, recs, out, m), "pt_debug_bounce")
        ref = orc.bounce(ps, recs)
        checked += m
        nxt = []
        for k in range(m):
            a, b = out[k], ref[k]
            same = (a.status == b.status and a.hittable == b.hittable and f32(a.t) == f32(b.t) and a.rng_state == b.rng_state
                    and a.front_face == b.front_face and list(a.sc_dir) == list(b.sc_dir) and list(a.sc_origin) == list(b.sc_origin)
                    and [f32(x) for x in a.color] == [f32(x) for x in b.color])
            if not same:
                bad.append(f"generation {g} ray {k}: origin {[float(np.float32(x)).hex() for x in recs[k].origin]} "
                           f"dir {[float(np.float32(x)).hex() for x in recs[k].dir]}: device status {a.status} hittable {a.hittable} "
                           f"t {a.t!r}, oracle status {b.status} hittable {b.hittable} t {b.t!r}")
            if b.status == abi.PT_BOUNCE_SCATTERED:
                nxt.append(k)
        if verbose:
            print(f"generation {g}: {m} rays, {len(bad)} mismatches so far, {len(nxt)} scattered")
        new = (abi.PtBounceIn * len(nxt))()
        for i, k in enumerate(nxt):
            new[i].origin[:] = list(ref[k].sc_origin); new[i].dir[:] = list(ref[k].sc_dir); new[i].time = ref[k].sc_time
            new[i].rng_state = ref[k].rng_state; new[i].attenuation[:] = list(ref[k].color)
        recs = new
    return checked, bad
