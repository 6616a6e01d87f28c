#!/bin/bash
# round 5, first GPU run of the rebuilt triangle pool: parity subset, timings, counters
mkdir -p gpurun_out/r05_tri
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "triangle or tri_pool or graze or triangles" 2>&1 | tail -15) > gpurun_out/r05_tri/tests.log
tail -5 gpurun_out/r05_tri/tests.log
(for a in "960 540 2" "1920 1080 8"; do timeout 300 python tools/tri_once.py $a; done) > gpurun_out/r05_tri/once.log 2>&1
cat gpurun_out/r05_tri/once.log
PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so timeout 300 python tools/tri_counters.py 2 960 540 > gpurun_out/r05_tri/counters.log 2>&1
cat gpurun_out/r05_tri/counters.log
