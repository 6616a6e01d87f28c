#!/bin/bash
# The evidence of round 6 on the final build, in one gpurun call (everything lands under gpurun_out/; afterwards, here: tools/collect_round6.sh).
#   gpurun --timeout 3400 -- 'bash tools/evidence_round6.sh [profiles-only]'
# Needs the diagnostic builds beside the shipped library:  make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_WALK   (libpt_stamps.so)
#                                                           make -C path_tracer_amd/csrc variant NAME=libpt_stamps_tri.so EXTRA="-DPT_STAMPS -DPT_STAMPS_TRI"
# `profiles-only`: the four profiled configs once more, to be run when profiles/ already holds PMC summaries of THIS build (the bench lines then
# carry the PMC-derived fields: bench.py nulls recordings of another build).
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f
mkdir -p $O
python tools/gpu_health.py 2>&1 | grep -v amdgpu | tee $O/gpu_health.txt
if [ "$1" != "profiles-only" ]; then
  python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
  tail -3 $O/gpu_tests.log
fi
tools/profile_round.sh r06_cornell 3 1 > $O/profile_cornell.log 2>&1                                               # cfg2, the headline
tools/profile_round.sh r06_smoke 3 1 --config cfg3 > $O/profile_smoke.log 2>&1                                      # cfg3
tools/profile_round.sh r06_cfg1 20 3 --config cfg1 > $O/profile_cfg1.log 2>&1                                       # cfg1: the reference's own workload
PT_PROFILE_MEM=1 tools/profile_round.sh r06_triangles 1 0 --config cfg5 > $O/profile_triangles.log 2>&1            # cfg5 (+ the memory-system passes)
python bench.py --steps 20 --warmup 2 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
if [ "$1" != "profiles-only" ]; then
  python bench.py --steps 5 --warmup 1 --mode fast > $O/bench_cfg2_fast_mode.json 2>/dev/null
  python bench.py --gpus 1 --dist-single --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_dist_single.json 2>/dev/null
  PT_ROUND=6 PT_SHARD_JSON=$O/shard_table_cornell_1080p_1024spp.json python tools/shard_table.py cornell 2>&1 | grep -v amdgpu.ids > $O/shard_table_cornell_1080p_1024spp.txt
  PT_ROUND=6 PT_SHARD_JSON=$O/shard_table_smoke_4k_512spp.json python tools/shard_table.py smoke 3840 2160 512 2>&1 | grep -v amdgpu.ids > $O/shard_table_smoke_4k_512spp.txt
  PT_ROUND=6 PT_SHARD_JSON=$O/shard_table_smoke_4k_4096spp.json python tools/shard_table.py smoke 3840 2160 4096 2>&1 | grep -v amdgpu.ids > $O/shard_table_smoke_4k_4096spp.txt
  PT_ROUND=6 PT_SHARD_JSON=$O/shard_table_triangles_1080p_64spp.json python tools/shard_table.py triangles 1920 1080 64 2>&1 | grep -v amdgpu.ids > $O/shard_table_triangles_1080p_64spp.txt
  # in-kernel counters of the two culling structures (what bench.py prices the culled algorithms with)
  [ -f path_tracer_amd/libpt_stamps.so ] && PT_FINAL_ROUND=6 PT_WALK_JSON=$O/smoke_walk_counters.json PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so \
    python tools/stamps.py smoke 128 0 lpt 2>&1 | grep -v amdgpu.ids > $O/smoke_walk_stamps.txt
  [ -f path_tracer_amd/libpt_stamps_tri.so ] && PT_FINAL_ROUND=6 PT_TRI_JSON=$O/tripool_counters.json PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps_tri.so \
    python tools/tri_counters.py 8 1920 1080 2>&1 | grep -v amdgpu.ids > $O/tripool_counters.txt
  # the path-ray soaks on the final build: triangle fields through the pool, everything else; the frame-level soak of the pool's three renderers
  (python tools/soak_path_rays.py 8 20000 triangle 2>&1 | grep -v amdgpu | tail -3) > $O/soak_triangle_fields.log
  (python tools/soak_path_rays.py 2 20000 box,sphere,random,random-img 2>&1 | grep -v amdgpu | tail -3) > $O/soak_all_kinds.log
  (python tools/soak_tri_renderers.py 200 66 2>&1 | grep -v amdgpu | tail -3) > $O/soak_tri_renderers_final.log
  (python tools/soak_scheduling.py 200 6 2>&1 | grep -v amdgpu | tail -3) > $O/soak_scheduling.log
fi
cat $O/*.txt | cut -c1-220
for t in r06_cornell r06_smoke r06_cfg1 r06_triangles; do cut -c1-320 gpurun_out/$t/bench_n1.json; done
cut -c1-320 $O/bench_cfg2_steps20.json
