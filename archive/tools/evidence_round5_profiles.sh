#!/bin/bash
# The four profiled configs of tools/evidence_round5.sh once more (bench line, kernel trace, PMC passes each on ONE box), to be run when
# profiles/ already holds PMC summaries of THIS build: the bench lines then carry the PMC-derived fields (bench.py nulls recordings of
# another build).  Afterwards, here: tools/collect_round5.sh.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f
mkdir -p $O
tools/profile_round.sh r05_cornell 3 1 > $O/profile_cornell.log 2>&1
tools/profile_round.sh r05_smoke 3 1 --config cfg3 > $O/profile_smoke.log 2>&1
tools/profile_round.sh r05_cfg1 20 3 --config cfg1 > $O/profile_cfg1.log 2>&1
PT_PROFILE_MEM=1 tools/profile_round.sh r05_triangles 1 0 --config cfg5 > $O/profile_triangles.log 2>&1
python bench.py --steps 20 --warmup 2 > $O/bench_cfg2_steps20.json 2> $O/bench_cfg2_steps20.err
python bench.py --steps 5 --warmup 1 --mode fast > $O/bench_cfg2_fast_mode.json 2>/dev/null
python bench.py --gpus 1 --dist-single --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg2_dist_single.json 2>/dev/null
for t in r05_cornell r05_smoke r05_cfg1 r05_triangles; do cut -c1-260 gpurun_out/$t/bench_n1.json; done
cut -c1-260 $O/bench_cfg2_steps20.json
