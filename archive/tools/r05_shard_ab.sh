#!/bin/bash
# A/B of library builds on shards of a frame (the chain floor), one gpurun call:
#   LIBS="libpt_render.so libpt_var_x.so" [SCENE=cornell ARGS="1920 1080 1024" SHARDS="1 8 16" REPS=2] tools/r05_shard_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_shard_ab
O=gpurun_out/r05_shard_ab/${TAG:-ab}.log
: > $O
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    echo "== $lib" >> $O
    PT_RENDER_LIB=$PWD/path_tracer_amd/$lib PT_RENDER_LIB_ALLOW_OLDER=1 SHARDS="${SHARDS:-1 8 16}" timeout 600 python tools/r05_prio.py ${SCENE:-cornell} ${ARGS:-1920 1080 1024} 0 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
