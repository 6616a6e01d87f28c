#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -q -x -k "tri or pool or mesh" 2>&1 | tail -2
python tools/soak_path_rays.py 2 20000 triangle 2>&1 | grep -v amdgpu | tail -3
