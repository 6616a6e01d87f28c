"""Register / scratch / occupancy table of the render kernels from csrc/build/resource_usage.txt (make -C path_tracer_amd/csrc asm)."""
import re
import sys
from pathlib import Path

t = (Path(__file__).resolve().parent.parent / "path_tracer_amd/csrc/build/resource_usage.txt").read_text()
pat = sys.argv[1] if len(sys.argv) > 1 else "render_kernel"
for b in t.split("Function Name: ")[1:]:
    name = b.split()[0]
    if pat not in name:
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    print(f"{name[17:-14]:64s} VGPR {g('VGPRs'):>3s} SGPR {g('SGPRs'):>3s} scratch {g('ScratchSize [bytes/lane]'):>3s} occ {g('Occupancy [waves/SIMD]')}")
