#!/bin/bash
# The evidence of round 3 on the final build, in one gpurun call (everything lands under gpurun_out/; afterwards, here:
#   for s in "r03_cornell cornell 1920 1080 1024" "r03_smoke smoke 1920 1080 1024" "r03_triangles triangles 1920 1080 256"; do
#     PT_FINAL_ROUND=3 tools/collect_profiles.sh $s; done         and copy gpurun_out/r03f/* to profiles/r03_*).
#   gpurun --timeout 4500 -- 'bash tools/evidence_round3.sh'
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q > gpurun_out/r03f/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gpu_tests.log
tail -3 gpurun_out/r03f/gpu_tests.log
tools/profile_round.sh r03_cornell 3 1 > gpurun_out/r03f/profile_cornell.log 2>&1                     # cfg2, the headline
tools/profile_round.sh r03_smoke 3 1 --config cfg3 > gpurun_out/r03f/profile_smoke.log 2>&1           # cfg3
[ -n "$PT_EVIDENCE_CFG5" ] && tools/profile_round.sh r03_triangles 1 0 --config cfg5 > gpurun_out/r03f/profile_triangles.log 2>&1   # cfg5: ~7 min
python bench.py --steps 20 --warmup 2 > gpurun_out/r03f/bench_cfg2_steps20.json 2> gpurun_out/r03f/bench_cfg2_steps20.err
python bench.py --config cfg3 --steps 3 --warmup 1 --width 400 --height 225 --spp 64 > gpurun_out/r03f/bench_cfg1_400x225x64.json 2>/dev/null
python bench.py --steps 5 --warmup 1 --mode fast > gpurun_out/r03f/bench_cfg2_fast_mode.json 2>/dev/null
PT_SHARD_JSON=gpurun_out/r03f/shard_table_cornell_1080p_1024spp.json python tools/shard_table.py cornell 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/shard_table_cornell_1080p_1024spp.txt
PT_SHARD_JSON=gpurun_out/r03f/shard_table_smoke_4k_512spp.json python tools/shard_table.py smoke 3840 2160 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/shard_table_smoke_4k_512spp.txt
# in-kernel counters of the sphere-grid walk (diagnostic build: make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_WALK)
[ -f path_tracer_amd/libpt_stamps.so ] && PT_WALK_JSON=gpurun_out/r03f/smoke_walk_counters.json PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so \
  python tools/stamps.py smoke 128 0 lpt 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/smoke_walk_stamps.txt
cat gpurun_out/r03f/*.txt
cut -c1-300 gpurun_out/r03_cornell/bench_n1.json gpurun_out/r03_smoke/bench_n1.json
