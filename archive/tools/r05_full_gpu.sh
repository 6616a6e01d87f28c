#!/bin/bash
# full GPU suite + cfg5 bench line
mkdir -p gpurun_out/r05_full
cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > gpurun_out/r05_full/gpu_tests.log
tail -5 gpurun_out/r05_full/gpu_tests.log
timeout 900 python bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_full/bench_cfg5.json 2> gpurun_out/r05_full/bench_cfg5.err
cat gpurun_out/r05_full/bench_cfg5.json | head -c 1500
