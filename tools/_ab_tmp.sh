#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -x -k "checker or cfg or smoke or mixed or golden" 2>&1 | tail -2
tools/abn.sh "libpt_var_nochk32.so libpt_render.so libpt_var_nochk32.so libpt_render.so" smoke 512 1 3840 2160
tools/abn.sh "libpt_var_nochk32.so libpt_render.so" smoke 256 8
