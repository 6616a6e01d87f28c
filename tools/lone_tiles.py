"""Lone-tile chain times: single 8x8 tiles of a frame rendered ALONE on the GPU (one wave: shard i of n_tiles), against the whole
frame.  max = the frame's makespan floor; sum / wave slots = what the frame would take if a wave's pace did not depend on its
neighbours (latency-bound); the frame's own time over that = how much the waves slow each other down (throughput-bound).
    python tools/lone_tiles.py [scene W H spp n_sample]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from path_tracer_amd import abi, render as R, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "smoke"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
n_sample = int(sys.argv[5]) if len(sys.argv) > 5 else 96
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
tx, ty = (W + 7) // 8, (H + 7) // 8
nt = tx * ty
R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
frame_ms = min(R.render(W, H, spp, ds, cam, timed=True)[1] for _ in range(2))
rng = np.random.default_rng(7)
tiles = sorted(set(rng.integers(0, nt, n_sample).tolist()))
ms = []
for t in tiles:
    R.render(W, H, 4, ds, cam, shard_index=t, shard_count=nt, flags=abi.PT_FLAG_NO_LPT)
    ms.append(min(R.render(W, H, spp, ds, cam, shard_index=t, shard_count=nt, flags=abi.PT_FLAG_NO_LPT, timed=True)[1] for _ in range(2)))
ms = np.array(ms)
print(f"{scene} {W}x{H}x{spp}: whole frame {frame_ms:.1f} ms, {nt} tiles")
print(f"lone tiles ({len(tiles)} sampled): mean {ms.mean():.2f} ms, median {np.median(ms):.2f}, p90 {np.percentile(ms, 90):.2f}, max {ms.max():.2f} (tile {tiles[int(ms.argmax())]})")
for slots in (4096, 5120):
    print(f"  sum over all tiles / {slots} wave slots = {ms.mean() * nt / slots:.1f} ms  -> the frame takes {frame_ms / (ms.mean() * nt / slots):.2f} x that")
order = np.argsort(-ms)[:8]
print("  heaviest sampled tiles (tile: ms):", ", ".join(f"{tiles[i]} ({tiles[i] % tx},{tiles[i] // tx}): {ms[i]:.1f}" for i in order))
