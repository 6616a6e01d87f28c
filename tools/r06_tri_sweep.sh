#!/bin/bash
# round 6: the pool's knobs again, now that camera rays take their candidates from the cache: SWEEP = "ENV=.. ENV=..;ENV=.." groups, 1080p x SPP
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/tri_sweep.log
: > $O
IFS=';' read -ra SWEEPS <<< "${SWEEP:-PT_TRI_M=8;PT_TRI_M=12;PT_TRI_M=16}"
for g in "${SWEEPS[@]}"; do
  echo "== $g" >> $O
  env $g timeout 900 python tools/tri_once.py 1920 1080 ${SPP:-32} 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
