import sys, zlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import scenes_small as S
from oracle import binding as orc
from test_gpu_parity import random_bounce_inputs
from path_tracer_amd import abi
from path_tracer_amd import render as R
name = sys.argv[1]
ps, _ = S.ALL[name]()
rng = np.random.default_rng(5)
n = 6000
recs = random_bounce_inputs(rng, n, np.float32((0, 0.3, -1)), np.float32(3.0))
orc.set_math(True)
ref = orc.bounce(ps, recs)
lib = abi.load_library()
ds = R.DeviceScene(ps)
out = (abi.PtBounceOut * n)()
abi.check(lib.pt_debug_bounce(ds.handle, recs, out, n), "x")
cnt = 0
for k in range(n):
    g, r = out[k], ref[k]
    if list(g.sc_origin) != list(r.sc_origin) or list(g.sc_dir) != list(r.sc_dir) or g.rng_state != r.rng_state:
        cnt += 1
        if cnt < 6:
            print(k, "status", g.status, r.status, "hit", g.hittable, r.hittable, "matkind", ps.materials[r.material].kind, "t", g.t, r.t)
            print("  p", list(g.p), list(r.p)); print("  sco", list(g.sc_origin), list(r.sc_origin)); print("  scd", list(g.sc_dir), list(r.sc_dir))
            print("  in o", list(recs[k].origin), "d", list(recs[k].dir), "tm", recs[k].time)
print("mismatching", cnt)
