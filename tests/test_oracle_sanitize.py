"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on this pool; the reference's own
CMake offers -fsanitize=thread for its host build, CMakeLists.txt:76-80).  Runs in a subprocess with libasan
preloaded, renders every small test scene once and checks the result against the uninstrumented build."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import scenes_small as S
from path_tracer_amd import abi, scenes
from oracle import binding as orc
san = C.CDLL(os.path.join(sys.argv[1], "oracle", "liboracle_asan.so"))
san.orc_render.argtypes = orc.load().orc_render.argtypes
san.orc_set_math.argtypes = [C.c_int]
for name, fn in S.ALL.items():
    ps, cam = fn()
    c = scenes.make_camera(cam, 20, 12)
    for mode in (0, 1):
        orc.set_math(bool(mode)); san.orc_set_math(mode)
        ref = orc.render(ps, c.c, 20, 12, 3)
        fb = np.zeros((12, 20, 3), np.float32)
        p = orc.params(20, 12, 3)
        rc = san.orc_render(C.byref(ps.desc), C.byref(c.c), C.byref(p), fb.ctypes.data_as(C.POINTER(C.c_float)), None)
        assert rc == 0
        assert fb.tobytes() == ref.tobytes(), name
    # the two extra executors of round 2: fast mode (own RNG streams per chunk) and the single-task stream
    orc.set_math(True); san.orc_set_math(1)
    for flags, spp in ((abi.PT_FLAG_FAST_RNG, 70), (abi.PT_FLAG_SINGLE_STREAM, 2)):
        ref = orc.render(ps, c.c, 20, 12, spp, flags=flags)
        fb = np.zeros((12, 20, 3), np.float32)
        p = orc.params(20, 12, spp, flags=flags)
        assert san.orc_render(C.byref(ps.desc), C.byref(c.c), C.byref(p), fb.ctypes.data_as(C.POINTER(C.c_float)), None) == 0
        assert fb.tobytes() == ref.tobytes(), (name, flags)
print("SANITIZED_OK")
'''


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "liboracle_asan.so"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", CHILD, str(ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZED_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
