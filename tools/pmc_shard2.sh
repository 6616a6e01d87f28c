#!/bin/bash
# tools/pmc_shard2.sh tag scene spp N "COUNTERS..."   one --pmc pass over one shard render
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $5 --output-format csv -d $OUT/c -- python3 tools/shard_once.py $2 $3 $4 > $OUT/c.log 2>&1
tail -1 $OUT/c.log
python3 tools/pmc_rows.py $OUT | grep -v rocclr
