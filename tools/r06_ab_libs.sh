#!/bin/bash
# round 6: builds side by side on one box: tools/r06_ab_libs.sh "lib1 lib2 ..." scene spp [steps] [extra bench args]
cd $GRAFT_REPO_ROOT
python tools/gpu_health.py 2>&1 | grep -v amdgpu
for rep in 1 2; do
for lib in $1; do
  echo "== $lib"
  PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python bench.py --steps ${4:-2} --warmup 1 --scene $2 --spp $3 --no-cpu-baseline $5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$2 spp=$3', d['value'], 'Msamples/s  kernel_ms', d['roofline']['kernel_ms'])"
done
done
