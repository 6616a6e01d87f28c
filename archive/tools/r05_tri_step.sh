#!/bin/bash
# round 5: one iteration of the pool's rebuild: quick parity subset + sweep + counters (knobs: $SWEEP = list of "ENV=.. ENV=.." groups separated by ;)
mkdir -p gpurun_out/r05_tri
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_tri/step.log
: > $O
[ -n "$SKIP_TESTS" ] || (timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_triangle_fields or graze" 2>&1 | tail -4) >> $O
run() { echo "== $*" >> $O; env "$@" timeout 600 python tools/tri_once.py 1920 1080 8 2>&1 | grep -v amdgpu.ids >> $O; }
IFS=';' read -ra SWEEPS <<< "${SWEEP:-PT_TRI_M=8;PT_TRI_M=16;PT_TRI_M=32}"
for g in "${SWEEPS[@]}"; do run $g; done
for m in ${COUNT_M:-16}; do
  echo "== counters PT_TRI_M=$m" >> $O
  PT_TRI_M=$m PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so timeout 300 python tools/tri_counters.py 2 960 540 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
