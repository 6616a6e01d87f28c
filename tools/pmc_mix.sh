#!/bin/bash
# VALU instruction mix of the render kernel by type (SQ_INSTS_VALU_*), per sample, for several builds, one gpurun call:
#   tools/pmc_mix.sh TAG "lib1 lib2 ..." SCENE SPP [W H]        -> gpurun_out/TAG/<lib>_mix.txt
TAG=$1; LIBS=$2; SCENE=$3; SPP=$4; W=${5:-1920}; H=${6:-1080}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64"
P2="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SALU SQ_INSTS_LDS"
for lib in $LIBS; do
  export PT_RENDER_LIB=$GRAFT_REPO_ROOT/path_tracer_amd/$lib PT_RENDER_LIB_ALLOW_OLDER=1
  n=${lib%.so}
  rocprofv3 --pmc $P1 --output-format csv -d $OUT/${n}_a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene $SCENE --spp $SPP --width $W --height $H > $OUT/${n}_a.log 2>&1
  rocprofv3 --pmc $P2 --output-format csv -d $OUT/${n}_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --scene $SCENE --spp $SPP --width $W --height $H > $OUT/${n}_b.log 2>&1
  python - $OUT $n $W $H $SPP <<'PY' | tee $OUT/${n}_mix.txt
import csv, glob, sys, collections
out, n, W, H, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
per = {}
for p in "ab":
    for f in glob.glob(f"{out}/{n}_{p}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
        dur = collections.defaultdict(float)
        for r in rows: dur[r["Dispatch_Id"]] = max(dur[r["Dispatch_Id"]], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        frame = max(dur, key=dur.get)
        for r in rows:
            if r["Dispatch_Id"] == frame: per[r["Counter_Name"]] = per.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
s = W * H * spp
print(f"== {n}: wave-instructions per sample, frame launch of {W}x{H}x{spp}")
for k in sorted(per): print(f"  {k:28s} {per[k] / s:9.3f}")
PY
  rm -rf $OUT/${n}_a $OUT/${n}_b
done
