#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
{
for lib in libpt_stamps0.so libpt_stamps2.so; do
  echo "== $lib"
  PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python tools/stamps.py smoke 128 0 lpt
  PT_STAMPS_WALK=1 PT_W=3840 PT_H=2160 PT_SHARDS=8 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python tools/stamps.py smoke 256 0 lpt
done
} > gpurun_out/r03a/stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/r03a/stamps.log
