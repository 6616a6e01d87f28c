#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
O=gpurun_out/r05_ab/sphere_nb.log
: > $O
(PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_var_nb.so PT_RENDER_LIB_ALLOW_OLDER=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "sphere or smoke or queued or cfg1 or cfg4 or mixed" 2>&1 | tail -4) >> $O
for rep in 1 2; do
  bash tools/abn.sh "libpt_render.so libpt_var_nb.so" smoke 256 1 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "libpt_render.so libpt_var_nb.so" smoke 256 8 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "libpt_render.so libpt_var_nb.so" smoke 64 1 400 225 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "libpt_render.so libpt_var_nb.so" smoke 128 8 3840 2160 2>&1 | grep -v amdgpu >> $O
done
cat $O
