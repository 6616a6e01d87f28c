#!/bin/bash
# Round-3 evidence on the final build (run through gpurun; everything lands in gpurun_out/, tools/collect_profiles.sh copies).
cd $GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q > gpurun_out/r03f/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gpu_tests.log
tail -3 gpurun_out/r03f/gpu_tests.log
tools/profile_round.sh r03_cornell 3 1 > gpurun_out/r03f/profile_cornell.log 2>&1
tools/profile_round.sh r03_smoke 3 1 --config cfg3 > gpurun_out/r03f/profile_smoke.log 2>&1
python bench.py --steps 20 --warmup 2 > gpurun_out/r03f/bench_cfg2_steps20.json 2> gpurun_out/r03f/bench_cfg2_steps20.err
python bench.py --config cfg3 --steps 3 --warmup 1 --width 400 --height 225 --spp 64 > gpurun_out/r03f/bench_cfg1_400x225x64.json 2>/dev/null
python bench.py --steps 5 --warmup 1 --mode fast > gpurun_out/r03f/bench_cfg2_fast_mode.json 2>/dev/null
PT_SHARD_JSON=gpurun_out/r03f/shard_table_cornell_1080p_1024spp.json python tools/shard_table.py cornell 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/shard_table_cornell_1080p_1024spp.txt
PT_SHARD_JSON=gpurun_out/r03f/shard_table_smoke_4k_512spp.json python tools/shard_table.py smoke 3840 2160 512 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/shard_table_smoke_4k_512spp.txt
PT_WALK_JSON=gpurun_out/r03f/smoke_walk_counters.json PT_STAMPS_WALK=1 PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_stamps.so python tools/stamps.py smoke 128 0 lpt 2>&1 | grep -v amdgpu.ids > gpurun_out/r03f/smoke_walk_stamps.txt
cat gpurun_out/r03f/*.txt
cat gpurun_out/r03_cornell/bench_n1.json gpurun_out/r03_smoke/bench_n1.json gpurun_out/r03f/bench_cfg1_400x225x64.json | cut -c1-400
