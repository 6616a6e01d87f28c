"""Regenerates the committed fixtures.  Run in the build container:  python tests/golden/make_golden.py

  xorshift32_kat.json      from oracle/_ref/xorshift_kat = the REFERENCE's own include/xorshift.hpp
                           compiled where it lies under /root/reference (oracle/Makefile target `ref`)
  fb_<scene>_32x18x4.npy   float framebuffers of the small test scenes from the CPU oracle in
                           portable-math mode (bit-comparable with the GPU; independent of the host libm)
"""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scenes_small as S  # noqa: E402
from oracle import binding as orc  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402

orc.build()
seeds = [1, 2, 3, 7, 12345, 2463534242, 4294967295, 2147483648, 1920 * 1080 - 1, 3840 * 2160 - 1]
out = subprocess.run([str(orc.REF_KAT), "32"] + [str(s) for s in seeds], capture_output=True, text=True, check=True)
streams = {}
for line in out.stdout.strip().splitlines():
    head, vals = line.split(":")
    streams[head.strip()] = [int(v) for v in vals.split()]
(HERE / "xorshift32_kat.json").write_text(json.dumps(
    {"source": "/root/reference/include/xorshift.hpp (xorshift<32>) via oracle/ref_xorshift_kat.cpp, g++ 11.4",
     "streams": streams}, indent=1))

orc.set_math(True)
for name, fn in S.ALL.items():
    ps, cam = fn()
    c = scenes.make_camera(cam, 32, 18)
    np.save(HERE / f"fb_{name}_32x18x4.npy", orc.render(ps, c.c, 32, 18, 4))
print("golden fixtures written to", HERE)
