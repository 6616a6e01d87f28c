#!/bin/bash
# Collects the evidence DESIGN.md / bench.py quote, on the GPU box:
#   tools/profile_round.sh TAG [STEPS WARMUP [bench.py args ...]]      e.g.  tools/profile_round.sh r02_smoke 3 1 --scene smoke
# bench line (N=1), rocprofv3 --kernel-trace --stats of the same command, and four --pmc passes (counters in their
# own runs, never combined with tracing).  Outputs land in gpurun_out/<tag>/; tools/collect_profiles.sh copies the
# summaries into profiles/.
TAG=${1:-r02}
STEPS=${2:-3}
WARMUP=${3:-1}
shift; shift; shift
ARGS="$@"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -c "from path_tracer_amd import abi; print(abi.load_library().pt_build_id().decode())" > $OUT/kernels_sha16.txt   # which build the counters below belong to: the LIBRARY that runs (pt_build_id), tools/pmc_summary.py
python bench.py --steps $STEPS --warmup $WARMUP $ARGS > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cat $OUT/bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps $STEPS --warmup $WARMUP --no-cpu-baseline $ARGS > $OUT/kt.log 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $P1 --output-format csv -d $OUT/pmc_a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $OUT/pmc_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_write.log 2>&1
if [ -n "$PT_PROFILE_MEM" ]; then  # the memory system behind the L1s: L2 hits / misses, what leaves L2 (fabric read requests, the part that goes to DRAM), L1 stalls
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_mem1.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mem2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_mem2.log 2>&1
fi
if [ -n "$PT_PROFILE_ICACHE" ]; then
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH --output-format csv -d $OUT/pmc_icache -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $ARGS > $OUT/pmc_icache.log 2>&1
fi
find $OUT -name "*.csv" | head -30
