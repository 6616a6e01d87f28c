"""The heaviest tiles of a chain-bound launch handed out 16 pixels at a time (PtTuning.heavy_tiles; pt_render.hip: launch / lane_acquire):
kernel ms of shard 0 of N without it (-1), with the launcher's rule (0) and with forced head lengths, and whether the image is the same.
    python tools/heavy_probe.py W H spp "N list" ["forced tile counts"]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes
W, H, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ns = [int(x) for x in sys.argv[4].split(",")]
forced = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else []
packed, cam_args = scenes.build("smoke")
cam = scenes.make_camera(cam_args, W, H)
print(f"smoke {W}x{H}x{spp}: kernel ms of shard 0/N by PtTuning.heavy_tiles (-1 never, 0 the launcher's rule, n forced)")
print("N    " + "".join(f"{t:>9d}" for t in [-1, 0] + forced) + "   images")
for n in ns:
    row, ref, same = [], None, True
    for t in [-1, 0] + forced:
        ds = R.DeviceScene(packed, tuning=abi.tuning(heavy_tiles=t))
        R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n)
        best, fb = 1e9, None
        for _ in range(3):
            fb, ms = R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)
            best = min(best, ms)
        if ref is None: ref = fb.clone()
        else: same = same and bool(torch.equal(ref.view(torch.int32), fb.view(torch.int32)))
        row.append(best)
    print(f"{n:<5d}" + "".join(f"{m:9.1f}" for m in row) + ("   identical" if same else "   DIFFERENT"), flush=True)
