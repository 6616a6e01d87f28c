#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "math_bit_exact or cfg or smoke or mixed or golden" 2>&1 | tail -2
tools/abn.sh "libpt_var_noballwave.so libpt_render.so libpt_var_noballwave.so libpt_render.so" smoke 512 1 3840 2160
tools/abn.sh "libpt_var_noballwave.so libpt_render.so" smoke 256 8
tools/abn.sh "libpt_var_noballwave.so libpt_render.so" smoke 256 1
tools/abn.sh "libpt_var_noballwave.so libpt_render.so" smoke 64 1 400 225
