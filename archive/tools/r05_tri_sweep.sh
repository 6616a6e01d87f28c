#!/bin/bash
# round 5: the rebuilt pool's parameter landscape (1080p x 8 spp, tools/tri_once.py)
mkdir -p gpurun_out/r05_tri
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tri/sweep.log
: > $O
run() { echo "== $*" >> $O; env "$@" timeout 600 python tools/tri_once.py 1920 1080 8 2>&1 | grep -v amdgpu.ids >> $O; }
run PT_TRI_M=8
run PT_TRI_M=16
run PT_TRI_M=32
run PT_TRI_M=64
run PT_TRI_M=16 PT_TRI_CELL=0.2
run PT_TRI_M=32 PT_TRI_CELL=0.2
run PT_TRI_M=64 PT_TRI_CELL=0.2
run PT_TRI_M=32 PT_TRI_CELL=0.2 PT_TRI_RES=256,64
run PT_TRI_M=8 PT_TRI_RES=256,64
cat $O
