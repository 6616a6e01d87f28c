#!/bin/bash
# The bench lines of the four profiled configs once more, with this round's PMC recordings and counters in the tree (the lines of
# tools/evidence_round5.sh were printed before profiles/r05_* existed: they quote round 4's recordings, nulled).  One gpurun call;
# afterwards, here:  for t in cornell smoke cfg1 triangles; do cp gpurun_out/r05_bench/${t}_bench_n1.json profiles/r05_${t}_bench_n1.json; done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bench
mkdir -p $O
python bench.py --steps 3 --warmup 1 > $O/cornell_bench_n1.json 2> $O/cornell.err
python bench.py --steps 3 --warmup 1 --config cfg3 > $O/smoke_bench_n1.json 2> $O/smoke.err
python bench.py --steps 20 --warmup 3 --config cfg1 > $O/cfg1_bench_n1.json 2> $O/cfg1.err
python bench.py --steps 1 --warmup 0 --config cfg5 > $O/triangles_bench_n1.json 2> $O/triangles.err
python bench.py --steps 20 --warmup 2 > $O/bench_cfg2_steps20.json 2> $O/steps20.err
for f in $O/*.json; do echo $f; cut -c1-400 $f; done
