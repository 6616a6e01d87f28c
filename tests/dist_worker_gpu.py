"""Worker of tests/test_gpu_distributed.py: one rank of a process group on the GPU box.

  mode "gloo":  world_size ranks share the one GPU of the box: each renders ITS shard with the HIP kernels
                (pt_render, shard_index = rank), the float tiles travel through a gloo gather (the box has one GPU, so
                RCCL cannot connect two ranks), and rank 0 un-interleaves on the GPU with pt_unshard_tiles.
  mode "nccl":  world size 1, backend nccl = RCCL: the product's exchange step exactly as `bench.py --gpus N` runs it
                (render of the rank's shard -> gather_frame on device tensors).
Rank 0 compares the assembled frame with the oracle, bit for bit."""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scenes_small as S  # noqa: E402
from oracle import binding as orc  # noqa: E402  (the checker)
from path_tracer_amd import render as R  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402


def main():
    mode, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    torch.cuda.set_device(0)
    if mode == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ps, cam = S.mixed_scene()
    c = scenes.make_camera(cam, w, h)
    if mode == "nccl":
        local = R.render(w, h, spp, ps, c, shard_index=rank, shard_count=world)
        frame = R.gather_frame(local, w, h)  # dist.gather of device tensors over RCCL, then the un-interleave (world 1: identity)
    else:
        local = R.render(w, h, spp, ps, c, shard_index=rank, shard_count=world)  # HIP-rendered shard
        torch.cuda.synchronize()
        frame = R.gather_frame(local.cpu(), w, h, None, lambda g, ww, hh, n: R.unshard(g.cuda(), ww, hh, n))
    if rank == 0:
        orc.set_math(True)
        full = orc.render(ps, c.c, w, h, spp)
        got = frame.cpu().numpy()
        same = (got.view(np.uint32) == full.view(np.uint32)) | (np.isnan(got) & np.isnan(full))
        assert same.all(), f"{int((~same).sum())} values differ"
        print("DIST_GPU_OK", mode, world, w, h)
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
