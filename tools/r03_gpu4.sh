#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
python -m pytest tests -m gpu -x -q > gpurun_out/r03c/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c/gpu_tests.log
tail -4 gpurun_out/r03c/gpu_tests.log
{
for i in 1 2; do
tools/abn.sh "libpt_r02.so" cornell 1024 1
PT_NO_MATSPEC=1 tools/abn.sh "libpt_render.so" cornell 1024 1
tools/abn.sh "libpt_render.so" cornell 1024 1
done
tools/abn.sh "libpt_r02.so libpt_render.so" cornell 1024 8
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r03c/ab.log
cat gpurun_out/r03c/ab.log
