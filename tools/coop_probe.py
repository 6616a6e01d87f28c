"""Chain-bound shards with the cooperative kernels forced: kernel ms of shard 0/N for flags 0 and PT_FLAG_FORCE_COOP, and a
PT_WIDE_LOGG / PT_SPLIT_TILES sweep under FORCE_COOP.   python tools/coop_probe.py scene N"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
scene, n = sys.argv[1], int(sys.argv[2])
W, H, SPP = 1920, 1080, 1024
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
def t(flags):
    d = R.DeviceScene(packed)
    R.render(W, H, 32, d, cam, flags=flags, shard_index=0, shard_count=n)
    return min(R.render(W, H, SPP, d, cam, flags=flags, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
for k in ("PT_SPLIT_TILES", "PT_WIDE_LOGG"): os.environ.pop(k, None)
print(f"{scene} shard 0/{n}: ordinary {t(0):.1f} ms, FORCE_COOP (model) {t(abi.PT_FLAG_FORCE_COOP):.1f} ms, FORCE_COOP|NO_SPLIT {t(abi.PT_FLAG_FORCE_COOP | abi.PT_FLAG_NO_SPLIT):.1f} ms", flush=True)
tiles = (W // 8) * (H // 8) // n
for lg in (1, 2, 3, 4):
    os.environ["PT_WIDE_LOGG"] = str(lg)
    row = []
    for frac in (0.005, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0):
        os.environ["PT_SPLIT_TILES"] = str(max(1, int(tiles * frac)))
        row.append(f"{frac*100:5.1f}%:{t(abi.PT_FLAG_FORCE_COOP):6.1f}")
    print(f"  G={1<<lg:2d}  " + "  ".join(row), flush=True)
