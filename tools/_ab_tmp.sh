#!/bin/bash
L="libpt_render.so libpt_var_sr2.so libpt_var_k4.so libpt_var_sr4.so"
tools/abn.sh "$L" smoke 512 1 3840 2160
tools/abn.sh "$L" smoke 512 1
tools/abn.sh "$L" smoke 256 8
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import torch
outs = {}
for lib in ("libpt_render.so", "libpt_var_sr2.so", "libpt_var_sr4.so", "libpt_var_k4.so"):
    import subprocess
    code = f"""
import os, sys
sys.path.insert(0, '.')
os.environ['PT_RENDER_LIB'] = os.getcwd() + '/path_tracer_amd/{lib}'; os.environ['PT_RENDER_LIB_ALLOW_OLDER'] = '1'
import torch, hashlib
from path_tracer_amd import render as R, scenes
packed, cam_args = scenes.build('smoke'); cam = scenes.make_camera(cam_args, 480, 270)
fb, _ = R.render(480, 270, 32, R.DeviceScene(packed), cam, timed=True)
print(hashlib.sha1(fb.cpu().numpy().tobytes()).hexdigest())
"""
    print(lib, subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip())
PY
