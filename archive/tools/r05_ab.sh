#!/bin/bash
# A/B of library builds on the 100 k-triangle mesh, one gpurun call:  LIBS="a.so b.so" [ARGS="1920 1080 8"] [REPS=2] tools/r05_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_tri
O=gpurun_out/r05_tri/ab.log
: > $O
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    echo "== $lib" >> $O
    PT_RENDER_LIB=$PWD/path_tracer_amd/$lib PT_RENDER_LIB_ALLOW_OLDER=1 timeout 600 python tools/tri_once.py ${ARGS:-1920 1080 8} 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
