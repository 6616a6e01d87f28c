#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/gpu_tests.log
tail -3 gpurun_out/r03a/gpu_tests.log
LIBS="libpt_r02.so libpt_render.so"
{
tools/abn.sh "$LIBS $LIBS" smoke 1024 1
tools/abn.sh "$LIBS" smoke 64 1 400 225
tools/abn.sh "$LIBS" smoke 512 8 3840 2160
tools/abn.sh "$LIBS" cornell 1024 1
PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so python tools/stamps.py smoke 128 0 lpt
} > gpurun_out/r03a/ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r03a/ab.log
