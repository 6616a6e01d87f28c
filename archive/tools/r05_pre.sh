#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
O=gpurun_out/r05_ab/prefilter.log
: > $O
(PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_var_pre.so PT_RENDER_LIB_ALLOW_OLDER=1 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_triangle_fields or graze or knobs" 2>&1 | tail -4) >> $O
for rep in 1 2 3; do
  for lib in libpt_render.so libpt_var_pre.so; do
    echo "== $lib" >> $O
    PT_RENDER_LIB=$PWD/path_tracer_amd/$lib PT_RENDER_LIB_ALLOW_OLDER=1 timeout 600 python tools/tri_once.py 1920 1080 8 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
