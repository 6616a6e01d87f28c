#!/bin/bash
cd $GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -q > gpurun_out/r03f/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03f/gpu_tests.log
tail -3 gpurun_out/r03f/gpu_tests.log
tools/profile_round.sh r03_cornell 3 1 > gpurun_out/r03f/profile_cornell.log 2>&1
tools/profile_round.sh r03_smoke 3 1 --config cfg3 > gpurun_out/r03f/profile_smoke.log 2>&1
python bench.py --steps 20 --warmup 2 > gpurun_out/r03f/bench_cfg2_steps20.json 2> gpurun_out/r03f/bench_cfg2_steps20.err
python bench.py --config cfg3 --steps 3 --warmup 1 --width 400 --height 225 --spp 64 > gpurun_out/r03f/bench_cfg1_400x225x64.json 2>/dev/null
tools/profile_round.sh r03_triangles 1 0 --config cfg5 > gpurun_out/r03f/profile_triangles.log 2>&1
cut -c1-300 gpurun_out/r03_cornell/bench_n1.json gpurun_out/r03_triangles/bench_n1.json
