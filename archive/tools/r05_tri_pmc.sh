#!/bin/bash
# round 5: PMC of the rebuilt pool, 1080p x 8 spp (env knobs pass through)
TAG=${1:-r05_tri_pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_a-z0-9]+|TCP_[A-Z_a-z0-9]+|SQ_INSTS_[A-Z_0-9]+|SQ_INST_LEVEL[A-Z_]+|SQ_WAIT[A-Z_]+)\b" | sort -u > $OUT/counters_list.txt
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P4="TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 tools/tri_once.py 1920 1080 8 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && (head -1 $f; grep render_kernel $f) > $OUT/pmc_$i.csv
  rm -rf $OUT/p$i
done
python - $OUT <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
per = {}
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    rows = list(csv.DictReader(open(f)))
    dur = collections.defaultdict(float)
    for r in rows: dur[r["Dispatch_Id"]] = max(dur[r["Dispatch_Id"]], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur: continue
    frame = max(dur, key=dur.get)
    here = {}
    for r in rows:
        if r["Dispatch_Id"] == frame: here[r["Counter_Name"]] = here.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta = [r for r in rows if r["Dispatch_Id"] == frame][0]
    print(f, "frame ms", dur[frame] / 1e6, "VGPR", meta["VGPR_Count"], "scratch", meta["Scratch_Size"], "LDS", meta["LDS_Block_Size"])
    for k, v in here.items(): per.setdefault(k, v)
for k in sorted(per): print(f"{k:36s} {per[k]:.4e}")
cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8
s = 1920 * 1080 * 8
if cyc:
    print("cycles/XCD", cyc, "ms", cyc / 2.4e6)
    print("VALU issue occupancy", per.get("SQ_INSTS_VALU", 0) * 2 / (1024 * cyc), "waves/SIMD", per.get("SQ_WAVE_CYCLES", 0) * 4 / (1024 * cyc))
    print("per sample wave-instr: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f SMEM %.0f" % tuple(per.get(k, 0) / s for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM")))
    if "SQ_THREAD_CYCLES_VALU" in per: print("lane utilisation", per["SQ_THREAD_CYCLES_VALU"] / (64 * per["SQ_ACTIVE_INST_VALU"]) if "SQ_ACTIVE_INST_VALU" in per else None)
    if "TCP_PENDING_STALL_CYCLES_sum" in per: print("L1 pending stall share", per["TCP_PENDING_STALL_CYCLES_sum"] / (256 * cyc))
    if "TCC_HIT_sum" in per: print("L2 hit rate", per["TCC_HIT_sum"] / max(1, per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), "bytes past L2", per.get("TCC_EA0_RDREQ_sum", 0) * 64)
    if "TA_TA_BUSY_sum" in per: print("TA busy share", per["TA_TA_BUSY_sum"] / (256 * cyc))
PY
