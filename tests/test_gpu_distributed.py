"""The N>1 path with HIP-rendered shards, on the one-GPU box: (a) 2 and 3 ranks share the GPU, each renders its own
shard with the kernels, the exchange step (render.gather_frame + pt_unshard_tiles) reassembles them — over gloo, because
RCCL cannot connect two ranks on one device; (b) the same step over the nccl backend (= RCCL) at world size 1, exactly the
code `bench.py --gpus N` runs.  Rank 0 checks the frame against the oracle bit for bit."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def run(world, mode, size, spp=4):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29900 + (os.getpid() % 300) + world + (7 if mode == "nccl" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "dist_worker_gpu.py"),
           mode, str(size[0]), str(size[1]), str(spp)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert f"DIST_GPU_OK {mode} {world} {size[0]} {size[1]}" in out.stdout


@pytest.mark.parametrize("world,size", [(2, (72, 40)), (3, (45, 21))])
def test_hip_shards_gathered_and_unsharded(world, size):
    run(world, "gloo", size)


def test_gather_frame_over_rccl_world_size_1():
    run(1, "nccl", (72, 40))
