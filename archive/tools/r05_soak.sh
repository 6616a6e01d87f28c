#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_soak
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1_5_million or full_size_triangle" -s 2>&1 | grep -v amdgpu.ids | tail -8) > gpurun_out/r05_soak/new_tests.log
cat gpurun_out/r05_soak/new_tests.log
(timeout 2400 python tools/soak_path_rays.py ${SOAK_MULT:-8} 20000 triangle 2>&1 | grep -v amdgpu | tail -5) > gpurun_out/r05_soak/soak_triangles.log
cat gpurun_out/r05_soak/soak_triangles.log
