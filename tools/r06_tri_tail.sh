#!/bin/bash
# round 6: where to hand the tail over (PT_BIN_TAIL = fraction of the pixels still live), 1080p x SPP
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/tri_tail.log
: > $O
if [ -n "$TESTS" ]; then (timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_triangle_fields or triangle_pool" 2>&1 | tail -5) >> $O; fi
for spp in ${SPPS:-8 32}; do
  for f in ${FRACS:-0 0.1 0.25 0.5 0.8}; do
    echo "== PT_BIN_TAIL=$f 1920x1080x$spp" >> $O
    PT_BIN_TAIL=$f timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
  done
  echo "== old 1920x1080x$spp" >> $O
  PT_TRI_UNBINNED=1 timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
