"""Register / scratch / LDS / occupancy of every kernel, from `make -C path_tracer_amd/csrc asm` (build/resource_usage.txt).
    python tools/resource_table.py [substring]"""
import re, subprocess, sys
from pathlib import Path
t = (Path(__file__).resolve().parent.parent / "path_tracer_amd/csrc/build/resource_usage.txt").read_text()
want = sys.argv[1] if len(sys.argv) > 1 else "render_kernel"
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
def g(b, k):
    m = re.search(re.escape(k) + r": (\d+)", b)
    return int(m.group(1)) if m else -1
for b in blocks:
    name = b.split("[")[0].split("\n")[0].strip()
    if want not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::KArgs)", "")
    print(f"{dem[:100]:100s} VGPR {g(b,'VGPRs'):3d} SGPR {g(b,'SGPRs'):3d} scratch {g(b,'ScratchSize [bytes/lane]'):4d} occ {g(b,'Occupancy [waves/SIMD]')} LDS {g(b,'LDS Size [bytes/block]')}")
