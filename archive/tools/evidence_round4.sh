#!/bin/bash
# The evidence of round 4 on the final build, in one gpurun call (everything lands under gpurun_out/; afterwards, here: tools/collect_round4.sh).
#   gpurun --timeout 3000 -- 'bash tools/evidence_round4.sh'
# Needs the diagnostic builds beside the shipped library:  make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_WALK   (libpt_stamps.so)
#                                                           make -C path_tracer_amd/csrc variant NAME=libpt_stamps_tri.so EXTRA="-DPT_STAMPS -DPT_STAMPS_TRI"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f
mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
tools/profile_round.sh r04_cornell 3 1 > $O/profile_cornell.log 2>&1                                               # cfg2, the headline
tools/profile_round.sh r04_smoke 3 1 --config cfg3 > $O/profile_smoke.log 2>&1                                      # cfg3
tools/profile_round.sh r04_cfg1 20 3 --config cfg1 > $O/profile_cfg1.log 2>&1    # cfg1: the reference's own workload
PT_PROFILE_MEM=1 tools/profile_round.sh r04_triangles 1 0 --config cfg5 > $O/profile_triangles.log 2>&1            # cfg5 (+ the memory-system passes)
python bench.py --steps 20 --warmup 2 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
python bench.py --steps 5 --warmup 1 --mode fast > $O/bench_cfg2_fast_mode.json 2>/dev/null
python bench.py --gpus 1 --dist-single --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_dist_single.json 2>/dev/null
PT_ROUND=4 PT_SHARD_JSON=$O/shard_table_cornell_1080p_1024spp.json python tools/shard_table.py cornell 2>&1 | grep -v amdgpu.ids > $O/shard_table_cornell_1080p_1024spp.txt
PT_ROUND=4 PT_SHARD_JSON=$O/shard_table_smoke_4k_512spp.json python tools/shard_table.py smoke 3840 2160 512 2>&1 | grep -v amdgpu.ids > $O/shard_table_smoke_4k_512spp.txt
PT_ROUND=4 PT_SHARD_JSON=$O/shard_table_smoke_4k_4096spp.json python tools/shard_table.py smoke 3840 2160 4096 2>&1 | grep -v amdgpu.ids > $O/shard_table_smoke_4k_4096spp.txt
PT_ROUND=4 PT_SHARD_JSON=$O/shard_table_triangles_1080p_64spp.json python tools/shard_table.py triangles 1920 1080 64 2>&1 | grep -v amdgpu.ids > $O/shard_table_triangles_1080p_64spp.txt
# in-kernel counters of the two culling structures (what bench.py prices the culled algorithms with)
[ -f path_tracer_amd/libpt_stamps.so ] && PT_FINAL_ROUND=4 PT_WALK_JSON=$O/smoke_walk_counters.json PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so \
  python tools/stamps.py smoke 128 0 lpt 2>&1 | grep -v amdgpu.ids > $O/smoke_walk_stamps.txt
[ -f path_tracer_amd/libpt_stamps_tri.so ] && PT_FINAL_ROUND=4 PT_TRI_JSON=$O/tripool_counters.json PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps_tri.so \
  python tools/tri_counters.py 2 960 540 2>&1 | grep -v amdgpu.ids > $O/tripool_counters.txt
python tools/lone_wave.py 2048 2>/dev/null > $O/lone_wave.txt
python tools/lone_tiles.py smoke 1920 1080 1024 128 2>/dev/null > $O/lone_tiles_smoke.txt
python tools/lone_tiles.py cornell 1920 1080 1024 96 2>/dev/null > $O/lone_tiles_cornell.txt
cat $O/*.txt | cut -c1-220
for t in r04_cornell r04_smoke r04_cfg1 r04_triangles; do cut -c1-260 gpurun_out/$t/bench_n1.json; done
