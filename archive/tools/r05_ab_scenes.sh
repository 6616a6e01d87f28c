#!/bin/bash
# A/B of library builds across the BASELINE scenes, one gpurun call (alternating, two rounds): LIBS="a.so b.so" tools/r05_ab_scenes.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
O=gpurun_out/r05_ab/scenes.log
: > $O
for rep in 1 2; do
  bash tools/abn.sh "$LIBS" smoke 256 1 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "$LIBS" smoke 256 8 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "$LIBS" smoke 64 1 400 225 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "$LIBS" cornell 1024 1 2>&1 | grep -v amdgpu >> $O
  bash tools/abn.sh "$LIBS" cornell 1024 8 2>&1 | grep -v amdgpu >> $O
done
cat $O
