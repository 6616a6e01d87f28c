#!/bin/bash
# round 6: builds of the binned renderer side by side (LIBS = files under path_tracer_amd/), 1080p x SPP; then the per-generation series of the first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/tri_ab.log
: > $O
if [ -n "$TESTS" ]; then (timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_triangle_fields or triangle_pool" 2>&1 | tail -5) >> $O; fi
for lib in ${LIBS:-libpt_render.so}; do
  for spp in ${SPPS:-8}; do
    echo "== $lib 1920x1080x$spp $ENVS" >> $O
    env $ENVS PT_RENDER_LIB=$PWD/path_tracer_amd/$lib timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
