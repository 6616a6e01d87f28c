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


def test_bench_distributed_line_schema():
    """`bench.py --gpus 1 --dist-single` runs the N > 1 code path (RCCL process group, sharded render, gather + un-interleave) at world
    size 1: the line must explain itself — every rank's kernel ms, the exchange step timed on its own, what the collective layer
    reports — so that a scaling curve measured on an 8-GPU node can be read without this repo's authors."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(29700 + os.getpid() % 200))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--dist-single", "--config", "cfg2", "--steps", "1", "--warmup", "0",
                          "--spp", "16", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.lstrip().startswith('{"metric"')][-1])
    d = line["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and len(d["kernel_ms_per_rank"]) == 1 and len(d["gather_unshard_ms_per_rank"]) == 1
    assert d["kernel_ms_per_rank"][0] > 0 and d["gather_unshard_ms_root"] >= 0 and d["rccl_version"]
    assert d["bytes_gathered_per_rank"] == 240 * 135 * 64 * 12
    assert "predicted_chain_floor_ms" in line and line["n_gpus"] == 1 and line["roofline"]["bound"] in ("valu", "hbm")
    assert line["roofline"]["hbm_algorithmic_bytes"] > 1920 * 1080 * 12  # the frame + the scene
