#!/bin/bash
tools/abn.sh "libpt_var_noud.so libpt_render.so libpt_var_noud.so libpt_render.so" smoke 512 1 3840 2160
tools/abn.sh "libpt_var_noud.so libpt_render.so" smoke 256 8
python -m pytest tests/test_gpu_parity.py -q -x -k "cfg or smoke or mixed or golden" 2>&1 | tail -2
