#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "triangle or graze" > gpurun_out/r03b/tri_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03b/tri_tests.log
tail -5 gpurun_out/r03b/tri_tests.log
{
for cfg in "16 1.0" "32 0.7" "24 0.8"; do
  set -- $cfg
  echo "== M=$1 cell=$2"
  PT_TRI_M=$1 PT_TRI_CELL=$2 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so python tools/tri_counters.py 2 480 270
  PT_TRI_M=$1 PT_TRI_CELL=$2 python tools/tri_once.py 960 540 8
done
PT_NO_TRICULL=1 python tools/tri_once.py 960 540 8
PT_TRI_M=16 PT_TRI_CELL=1.0 python tools/tri_once.py 1920 1080 8
PT_NO_TRICULL=1 python tools/tri_once.py 1920 1080 8
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03b/sweep3.log
cat gpurun_out/r03b/sweep3.log
