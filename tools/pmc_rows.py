"""Prints per-kernel counter values (largest dispatch per kernel) from rocprofv3 --pmc csv output directories."""
import csv, sys
from pathlib import Path
rows = {}
for f in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        v = float(r["Counter_Value"])
        key = (k, r["Counter_Name"])
        rows[key] = max(rows.get(key, 0.0), v)
for (k, c), v in sorted(rows.items()):
    print(f"{k:42s} {c:26s} {v:.4g}")
