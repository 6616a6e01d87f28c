#!/bin/bash
# Collects the evidence DESIGN.md / bench.py quote, on the GPU box:  tools/profile_round.sh r01b
#   bench line (N=1 default config), rocprofv3 --kernel-trace --stats of the same command, and four --pmc passes
#   (counters in their own runs, never combined with tracing).  Outputs land in gpurun_out/<tag>/.
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 1 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cat $OUT/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc_a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
