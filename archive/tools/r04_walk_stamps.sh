#!/bin/bash
mkdir -p gpurun_out/r04_walk
for v in old new; do
  echo "== $v" | tee -a gpurun_out/r04_walk/stamps.txt
  PT_STAMPS_WALK=1 PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps_$v.so python tools/stamps.py smoke 128 2>/dev/null | tee -a gpurun_out/r04_walk/stamps.txt
done
for lib in libpt_var_old.so libpt_render.so; do
  echo "== $lib" | tee -a gpurun_out/r04_walk/lone_wave.txt
  PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python tools/lone_wave.py 2048 2>/dev/null | head -1 | tee -a gpurun_out/r04_walk/lone_wave.txt
done
