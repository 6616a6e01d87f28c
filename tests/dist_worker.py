"""Worker of tests/test_distributed_cpu.py: one rank of a gloo process group on CPU.  Each rank produces its
shard of the frame (with the CPU oracle standing in for the GPU kernel, same shard layout), then runs the
product's exchange step path_tracer_amd.render.gather_frame; rank 0 checks the assembled frame."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scenes_small as S  # noqa: E402
from dist_util import unshard_reference  # noqa: E402
from oracle import binding as orc  # noqa: E402
from path_tracer_amd import render as R  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402


def main():
    w, h, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ps, cam = S.mixed_scene()
    c = scenes.make_camera(cam, w, h)
    orc.set_math(True)
    local = torch.from_numpy(orc.render(ps, c.c, w, h, spp, shard_index=rank, shard_count=world))

    def unshard_np(gathered, width, height, n):
        return torch.from_numpy(unshard_reference(gathered.numpy(), width, height, n))

    frame = R.gather_frame(local, w, h, None, unshard_np)
    if rank == 0:
        full = orc.render(ps, c.c, w, h, spp)
        got = frame.numpy()
        same = (got.view(np.uint32) == full.view(np.uint32)) | (np.isnan(got) & np.isnan(full))
        assert same.all(), f"{int((~same).sum())} values differ"
        print("DIST_OK", world, w, h)
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
