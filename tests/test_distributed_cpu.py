"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the product's exchange step
(render.gather_frame: one gather of the float tiles to rank 0 + un-interleave) on shards with the product's
tile layout; the assembled frame must equal the single-process frame bit for bit."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world,size", [(2, (40, 24)), (2, (21, 13)), (3, (40, 24))])
def test_gloo_gather_reassembles_the_frame(world, size):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    port = 29500 + (os.getpid() % 400) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "tests" / "dist_worker.py"),
           str(size[0]), str(size[1]), "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"DIST_OK {world} {size[0]} {size[1]}" in out.stdout
