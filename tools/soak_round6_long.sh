#!/bin/bash
# A longer soak on the final build with seeds the evidence pass did not use (leftover GPU minutes of round 6):
#   gpurun --timeout 3000 -- 'bash tools/soak_round6_long.sh'      -> gpurun_out/r06_long/*.log (copied to profiles/r06_soak_long_*.log)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_long
mkdir -p $O
python tools/gpu_health.py 2>&1 | grep -v amdgpu | tee $O/gpu_health.txt
python -c "from path_tracer_amd import abi; import ctypes as C; l = abi.load_library(); l.pt_build_id.restype = C.c_char_p; print('build id', l.pt_build_id().decode())" | tee $O/build_id.txt
(timeout 900 python tools/soak_tri_renderers.py ${N_TRI:-700} ${SEED_TRI:-101} 2>&1 | grep -v amdgpu | tail -2) > $O/tri_renderers.log
(timeout 600 python tools/soak_scheduling.py ${N_SCHED:-600} ${SEED_SCHED:-11} 2>&1 | grep -v amdgpu | tail -2) > $O/scheduling.log
(timeout 1500 python tools/soak_path_rays.py ${MULT_TRI:-10} 20000 triangle 2>&1 | grep -v amdgpu | tail -2) > $O/triangle_fields.log
(timeout 700 python tools/soak_path_rays.py ${MULT_ALL:-6} 20000 box,sphere,random,random-img 2>&1 | grep -v amdgpu | tail -2) > $O/all_kinds.log
python tools/gpu_health.py 2>&1 | grep -v amdgpu | tee -a $O/gpu_health.txt
tail -n 3 $O/*.log
