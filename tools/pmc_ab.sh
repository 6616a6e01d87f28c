#!/bin/bash
# PMC passes A and B for several builds of the extension, one gpurun call:
#   tools/pmc_ab.sh TAG "lib1 lib2 ..." SCENE SPP [W H] [extra bench args]   -> gpurun_out/TAG/<lib>_pmc_summary.json
TAG=$1; LIBS=$2; SCENE=$3; SPP=$4; W=${5:-1920}; H=${6:-1080}; shift 6
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
for lib in $LIBS; do
  export PT_RENDER_LIB=$GRAFT_REPO_ROOT/path_tracer_amd/$lib PT_RENDER_LIB_ALLOW_OLDER=1
  n=${lib%.so}
  rocprofv3 --pmc $P1 --output-format csv -d $OUT/${n}_a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene $SCENE --spp $SPP --width $W --height $H "$@" > $OUT/${n}_a.log 2>&1
  rocprofv3 --pmc $P2 --output-format csv -d $OUT/${n}_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene $SCENE --spp $SPP --width $W --height $H "$@" > $OUT/${n}_b.log 2>&1
  fa=$(find $OUT/${n}_a -name "*counter_collection.csv" | head -1); fb=$(find $OUT/${n}_b -name "*counter_collection.csv" | head -1)
  (head -1 $fa; grep render_kernel $fa) > $OUT/${n}_pmc_a.csv; (head -1 $fb; grep render_kernel $fb) > $OUT/${n}_pmc_b.csv
  python tools/pmc_summary.py $n $SCENE $W $H $SPP $OUT/${n}_pmc_a.csv $OUT/${n}_pmc_b.csv > $OUT/${n}_pmc_summary.json
  rm -rf $OUT/${n}_a $OUT/${n}_b
  echo "== $lib"; python -c "
import json; d=json.load(open('$OUT/${n}_pmc_summary.json')); p=d['per_launch']; s=$W*$H*$SPP
print({k: round(v,4) for k,v in d['derived'].items()})
print('per sample: VALU %.1f SALU %.1f LDS %.2f TRANS %.2f wave-instr; bank-conflict cycles/LDS instr %.2f; ms %.1f' % (p['SQ_INSTS_VALU']/s*1, p['SQ_INSTS_SALU']/s, p['SQ_INSTS_LDS']/s, p.get('SQ_INSTS_VALU_TRANS',0)/s, p['SQ_LDS_BANK_CONFLICT']/max(p['SQ_INSTS_LDS'],1), d['derived']['cycles_per_xcd']/2.4e6))
print(d['kernel'])"
done
