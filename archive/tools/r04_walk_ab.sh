#!/bin/bash
# round 4, sphere-grid walk: correctness of the queued walk + A/B of its variants + stamps + PMC, one gpurun call
#   tools/r04_walk_ab.sh "lib1 lib2 ..."     (libpt_stamps_new.so / libpt_stamps_old.so are used when present)
LIBS=${1:-"libpt_var_old.so libpt_render.so"}
mkdir -p gpurun_out/r04_walk
L=gpurun_out/r04_walk
python -m pytest tests/test_gpu_parity.py -x -q -k "sphere_grid or smoke or config1 or sphere_runs or image_texture or cfg1 or bounce_bit_exact or framebuffer_bit_exact" > $L/tests_parity.log 2>&1; echo "parity rc=$?" | tee -a $L/summary.txt
tail -3 $L/tests_parity.log
python -m pytest tests/test_gpu_fuzz.py -x -q -k "sphere" > $L/tests_fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $L/summary.txt
tail -3 $L/tests_fuzz.log
tools/abn.sh "$LIBS" smoke 1024 1 2>/dev/null | tee $L/ab_1080p.txt
tools/abn.sh "$LIBS" smoke 64 1 400 225 2>/dev/null | tee $L/ab_cfg1.txt
tools/abn.sh "$LIBS" smoke 512 8 3840 2160 2>/dev/null | tee $L/ab_4k_shard8.txt
for v in old new; do
  [ -f path_tracer_amd/libpt_stamps_$v.so ] || continue
  echo "== $v" | tee -a $L/stamps.txt
  PT_STAMPS_WALK=1 PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps_$v.so python tools/stamps.py smoke 128 2>/dev/null | tee -a $L/stamps.txt
done
for lib in $LIBS; do
  echo "== $lib" | tee -a $L/lone_wave.txt
  PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python tools/lone_wave.py 2048 2>/dev/null | head -1 | tee -a $L/lone_wave.txt
done
bash tools/pmc_ab.sh r04_walk_pmc "$LIBS" smoke 256 1920 1080 2>/dev/null
