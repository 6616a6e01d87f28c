#!/bin/bash
# Several builds side by side in ONE gpurun call:  tools/abn.sh "lib1 lib2 ..." scene spp shards [width height]
# (libs = files under path_tracer_amd/, e.g. libpt_render.so libpt_var_k2.so); prints kernel ms of shard 0 of N
for lib in $1; do
  PT_RENDER_LIB_ALLOW_OLDER=1 PT_RENDER_LIB=$PWD/path_tracer_amd/$lib python - "$2" "$3" "$4" "${5:-1920}" "${6:-1080}" "$lib" <<'PY'
import sys
sys.path.insert(0, '.')
import torch
from path_tracer_amd import render as R, scenes
scene, spp, n, W, H, lib = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 16, ds, cam, shard_index=0, shard_count=n)
ms = [R.render(W, H, spp, ds, cam, shard_index=0, shard_count=n, timed=True)[1] for _ in range(3)]
samples = W * H * spp / n
print(f"{lib:24s} {scene} {W}x{H}x{spp} shard 0/{n}: {min(ms):8.1f} ms  ({samples / min(ms) / 1e3:8.1f} Msamples/s)  all {[round(m, 1) for m in ms]}", flush=True)
PY
done
