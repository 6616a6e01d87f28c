#!/bin/bash
# Copies what tools/profile_round.sh left in gpurun_out/<tag>/ into profiles/ (tracked):
#   [PT_FINAL_ROUND=3] tools/collect_profiles.sh TAG SCENE W H SPP       e.g.  tools/collect_profiles.sh r02_smoke smoke 1920 1080 1024
# PT_FINAL_ROUND marks the PMC summary as the round's final one for its workload (what bench.py's pmc_traffic selects)
TAG=$1; SCENE=$2; W=$3; H=$4; SPP=$5
SRC=gpurun_out/$TAG
DST=profiles
cp $SRC/bench_n1.json $DST/${TAG}_bench_n1.json
cp $(ls -t $(find $SRC/kt -name "*kernel_stats.csv") | head -1) $DST/${TAG}_kernel_stats.csv   # the newest run of this tag
KT=$(ls -t $(find $SRC/kt -name "*kernel_trace.csv") | head -1)
(head -1 $KT; grep -E "render_kernel|lpt_order" $KT | head -8) > $DST/${TAG}_kernel_trace_rows.csv
for p in a b fetch write icache mem1 mem2; do
  [ -d $SRC/pmc_$p ] || continue
  f=$(ls -t $(find $SRC/pmc_$p -name "*counter_collection.csv" 2>/dev/null) 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "render_kernel" $f) > $DST/${TAG}_pmc_$p.csv
done
[ -f $SRC/kernels_sha16.txt ] && export PT_KERNELS_SHA16=$(cat $SRC/kernels_sha16.txt)
python tools/pmc_summary.py ${PT_FINAL_ROUND:+--final $PT_FINAL_ROUND} $TAG $SCENE $W $H $SPP $DST/${TAG}_pmc_a.csv $DST/${TAG}_pmc_b.csv $DST/${TAG}_pmc_fetch.csv $DST/${TAG}_pmc_write.csv $(ls $DST/${TAG}_pmc_mem1.csv $DST/${TAG}_pmc_mem2.csv 2>/dev/null) > $DST/${TAG}_pmc_summary.json
cat $DST/${TAG}_pmc_summary.json | python -c "import json,sys; d=json.load(sys.stdin); print(json.dumps(d['derived'], indent=1)); print(d['kernel'])"
