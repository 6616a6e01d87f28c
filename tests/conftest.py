import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def lib():
    """The product's HIP extension through its C ABI.  Built on demand (hipcc cross-compiles without a GPU); there
    is no fallback: if it cannot be built or loaded the tests fail."""
    from path_tracer_amd import abi
    if not abi.library_path().exists():
        import __graft_entry__
        __graft_entry__.build()
    return abi.load_library()


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_identical(a: np.ndarray, b: np.ndarray, what: str = ""):
    """Bit-exact float32 comparison; any NaN matches any NaN (payload/sign of a NaN is not semantics)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        idx = np.argwhere(~same)
        first = tuple(idx[0])
        raise AssertionError(f"{what}: {len(idx)} of {a.size} values differ; first at {first}: "
                             f"{a[first]!r} ({bits(a)[first]:#010x}) vs {b[first]!r} ({bits(b)[first]:#010x})")


def psnr_8bit(a8: np.ndarray, b8: np.ndarray) -> float:
    mse = np.mean((a8.astype(np.float64) - b8.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)
