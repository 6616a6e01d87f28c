#!/bin/bash
for r in 1 2; do
tools/abn.sh "libpt_var_nobehind.so libpt_render.so" smoke 256 1
done
tools/abn.sh "libpt_var_nobehind.so libpt_render.so" smoke 256 8
tools/abn.sh "libpt_var_nobehind.so libpt_render.so" smoke 512 1 3840 2160
tools/abn.sh "libpt_var_nobehind.so libpt_render.so" smoke 64 1 400 225
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "sphere or smoke or grid or cfg" 2>&1 | tail -2
