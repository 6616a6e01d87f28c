#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "math_bit_exact" 2>&1 | tail -2
for r in 1 2; do
tools/abn.sh "libpt_var_nosincos.so libpt_render.so" smoke 256 1
done
tools/abn.sh "libpt_var_nosincos.so libpt_render.so" smoke 256 8
tools/abn.sh "libpt_var_nosincos.so libpt_render.so" smoke 512 1 3840 2160
tools/abn.sh "libpt_var_nosincos.so libpt_render.so" smoke 64 1 400 225
