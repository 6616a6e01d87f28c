#!/bin/bash
# tools/pmc_shard3.sh tag scene spp N flags "COUNTERS..."   (env PT_* knobs pass through)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $6 --output-format csv -d $OUT/c -- python3 tools/shard_once.py $2 $3 $4 $5 > $OUT/c.log 2>&1
tail -1 $OUT/c.log
python3 tools/pmc_rows.py $OUT | grep -v rocclr | grep "^void"
