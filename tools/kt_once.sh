#!/bin/bash
# kernel trace of one shard render: tools/kt_once.sh tag scene spp N [flags]
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 tools/shard_once.py $2 $3 $4 ${5:-0} > $OUT/kt.log 2>&1
tail -1 $OUT/kt.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            print(r["Kernel_Name"][:60], "grid", r["Grid_Size"], "wg", r["Workgroup_Size"], "lds", r["LDS_Block_Size"], "vgpr", r["VGPR_Count"], "scratch", r["Scratch_Size"], "ms", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
PY
