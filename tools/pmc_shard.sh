#!/bin/bash
# PMC passes over one render of shard 0/N: tools/pmc_shard.sh tag scene spp N
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"
rocprofv3 --pmc $P1 --output-format csv -d $OUT/a -- python3 tools/shard_once.py $2 $3 $4 > $OUT/a.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $OUT/b -- python3 tools/shard_once.py $2 $3 $4 > $OUT/b.log 2>&1
tail -1 $OUT/a.log
python3 tools/pmc_rows.py $OUT
