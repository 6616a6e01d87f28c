"""Kernel ms of shard 0 of N for several caps on resident workgroups per CU (PtTuning.blocks_per_cu): how many waves should share a SIMD when a
launch has few tiles per wave slot.   python tools/blocks_sweep.py scene spp "1,2,4,8" "2,3,4,6,8" [W H]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
scene, spp = sys.argv[1], int(sys.argv[2])
ns = [int(x) for x in sys.argv[3].split(",")]
bs = [int(x) for x in sys.argv[4].split(",")]
W, H = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080)
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
print(f"{scene} {W}x{H}x{spp}: kernel ms of shard 0/N by blocks_per_cu (0 = the launcher's own choice)")
print("N    " + "".join(f"{b:>9d}" for b in [0] + bs))
for n in ns:
    row = []
    for b in [0] + bs:
        ds = R.DeviceScene(packed, tuning=abi.tuning(blocks_per_cu=b) if b else None)
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n)
        row.append(min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(3)))
    print(f"{n:<5d}" + "".join(f"{m:9.1f}" for m in row), flush=True)
