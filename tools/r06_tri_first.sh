#!/bin/bash
# round 6: the binned triangle-pool renderer, first contact: parity subset, then old (PT_TRI_UNBINNED) against new at 1080p x 8 / 32 spp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/tri_first.log
: > $O
(timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_triangle_fields or triangle_pool" 2>&1 | tail -15) >> $O
for spp in 8 32; do
  for mode in new old; do
    echo "== $mode 1920x1080x$spp" >> $O
    if [ $mode = old ]; then export PT_TRI_UNBINNED=1; else unset PT_TRI_UNBINNED; fi
    timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
    timeout 900 python tools/tri_once.py 1920 1080 $spp 2>&1 | grep -v amdgpu.ids >> $O
  done
done
unset PT_TRI_UNBINNED
cat $O
