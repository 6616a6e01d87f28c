"""What each rank of an N-GPU job runs, measured on ONE GPU: kernel time of shard 0 of N (N = 1, 2, 4, 8) of a frame, in
parity mode and in the opt-in fast mode (PT_FLAG_FAST_RNG).  The slowest rank sets the step time, shard 0 stands for it
(tiles are dealt round-robin, so the shards are statistically alike); the RCCL gather of the tiles is not included
(25 MB / N per rank at 1080p).  Predicted speed-up = t(1) / t(N).
    python tools/shard_table.py [cornell|smoke] [width height spp]
PT_SHARD_JSON=path: also writes {scene, workload, parity: {N: ms}, fast: {N: ms}} there (profiles/<tag>_shard_table_*.json: bench.py's
N > 1 line quotes the parity figure of its N as `predicted_chain_floor_ms`)."""
import json
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, render as R, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, SPP = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 1024)
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam); torch.cuda.synchronize()
print(f"{scene} {W}x{H}x{SPP}: kernel ms of shard 0 of N (one GPU), predicted N-GPU speed-up = t(1)/t(N)", flush=True)
record = {"scene": scene, "workload": f"{W}x{H}x{SPP}", "final": True, "round": int(os.environ.get("PT_ROUND", "3")),
          "note": "kernel ms of shard 0 of N measured on ONE GPU: what each rank of an N-GPU job runs (tools/shard_table.py)"}
for mode, flags in (("parity", 0), ("fast (PT_FLAG_FAST_RNG, not the reference's image)", abi.PT_FLAG_FAST_RNG)):
    t = {}
    for n in (1, 2, 4, 8):
        t[n] = min(R.render(W, H, SPP, ds, cam, flags=flags, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2 if n == 1 else 3))
    print(f"  {mode}:", "  ".join(f"N={n}: {t[n]:8.1f} ms ({t[1] / t[n]:4.2f}x)" for n in t), flush=True)
    record["parity" if flags == 0 else "fast"] = {str(n): round(t[n], 2) for n in t}
if os.environ.get("PT_SHARD_JSON"):
    json.dump(record, open(os.environ["PT_SHARD_JSON"], "w"), indent=1)
