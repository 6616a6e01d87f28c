#!/bin/bash
# round 6: per-kernel times of the binned renderer (rocprofv3 --kernel-trace --stats), 1080p x 8 spp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
rm -rf /tmp/prof_tri
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tri -o tri -- python3 tools/tri_once.py 1920 1080 ${SPP:-8} > gpurun_out/r06/tri_prof.log 2>&1
f=$(find /tmp/prof_tri -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r06/tri_kernel_stats.csv
cat gpurun_out/r06/tri_kernel_stats.csv | cut -c1-200
tail -3 gpurun_out/r06/tri_prof.log
t=$(find /tmp/prof_tri -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY' > gpurun_out/r06/tri_trace_series.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "bin_step" in r["Kernel_Name"]]
band = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "band_kernel" in r["Kernel_Name"]]
n = len(step) // 2
print("generations per render", n)
for g in range(n, 2 * n):
    print(g - n, f"step {step[g]:.0f} us  band {band[g]:.0f} us")
PY
head -60 gpurun_out/r06/tri_trace_series.txt
