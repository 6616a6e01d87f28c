"""Kernel time vs resident workgroups per CU (PT_BLOCKS_PER_CU), whole frame and shard 0 of 8."""
import os, sys, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import torch
    from path_tracer_amd import abi, scenes
    from path_tracer_amd import render as R
    scene, spp = sys.argv[2], int(sys.argv[3])
    W, H = 1920, 1080
    packed, cam_args = scenes.build(scene)
    cam = scenes.make_camera(cam_args, W, H)
    ds = R.DeviceScene(packed)
    R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
    out = []
    for n in (1, 2, 4, 8):
        ms = min(R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(2))
        out.append(f"1/{n}: {ms:7.1f} ms")
    print(f"{scene} blocks/CU={os.environ.get('PT_BLOCKS_PER_CU','max'):>3s}  " + "  ".join(out), flush=True)
else:
    scene, spp = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("cornell", "1024")
    for k in ("1", "2", "3", "4", "6"):
        subprocess.run([sys.executable, __file__, "child", scene, spp], env=dict(os.environ, PT_BLOCKS_PER_CU=k), stderr=subprocess.DEVNULL)
