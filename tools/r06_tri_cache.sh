#!/bin/bash
# round 6: the camera rays' candidate cache of the triangle-pool kernels, with / without (PT_NO_TRI_CACHE), builds LIBS, 1080p x SPPS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/tri_cache.log
: > $O
if [ -n "$TESTS" ]; then (timeout 1800 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "triangle" 2>&1 | tail -5) >> $O; fi
for spp in ${SPPS:-8 32}; do
  for lib in ${LIBS:-libpt_render.so}; do
    for e in ${ENVS:-PT_X=1 PT_NO_TRI_CACHE=1}; do
      echo "== $lib $e 1920x1080x$spp" >> $O
      env $e PT_RENDER_LIB=$PWD/path_tracer_amd/$lib timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
    done
  done
done
cat $O
