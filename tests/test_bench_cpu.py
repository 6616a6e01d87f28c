"""bench.py's recorded algorithmic op counts (used by ranks that do not run the cpu_baseline leg) agree with the
oracle's event counters, and the CLI contract holds."""
import importlib.util
import json
from pathlib import Path

import pytest

from path_tracer_amd import scenes

ROOT = Path(__file__).resolve().parent.parent


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("scene,w,h,spp", [("cornell", 480, 270, 4), ("smoke", 240, 135, 2)])
def test_recorded_ops_per_sample_match_oracle_counters(orc, scene, w, h, spp):
    bench = load_bench()
    packed, cam_args = scenes.build(scene)
    orc.set_math(True)
    _, ctr = orc.render(packed, scenes.make_camera(cam_args, w, h).c, w, h, spp, 50, counters=True)
    live = bench.ops_per_sample(ctr.as_dict())
    assert abs(live / bench.ALGORITHMIC_OPS_PER_SAMPLE[scene] - 1) < 0.08  # a smaller sample than the recorded one: Monte-Carlo noise


def test_peak_and_defaults():
    bench = load_bench()
    assert abs(bench.PEAK_TLANEOPS - 78.6) < 0.1  # 256 CU x 4 SIMD-32 x 2.4 GHz
    text = (ROOT / "bench.py").read_text()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in text
    # the committed bench lines parse and carry the two extra objects
    for f in sorted((ROOT / "profiles").glob("r*_bench_n1.json")):
        d = json.loads(f.read_text())
        assert d["unit"] == "Msamples/s" and "roofline" in d and "cpu_baseline" in d
        assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1


def test_weak_scaling_frames_keep_pixels_per_gpu_and_aspect():
    """bench.py --gpus N (weak scaling): N x the pixels of the 1080p frame, same aspect; N = 4 is exactly 4K."""
    import bench
    assert bench.weak_frame(1920, 1080, 1) == (1920, 1080)
    assert bench.weak_frame(1920, 1080, 4) == (3840, 2160)
    for n in (2, 3, 8):
        w, h = bench.weak_frame(1920, 1080, n)
        assert abs(w * h / (1920 * 1080 * n) - 1.0) < 2e-3
        assert abs(w / h - 1920 / 1080) < 2e-3
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        assert abs(tiles / n / 32400 - 1.0) < 0.01  # one 1-GPU frame's worth of tiles per rank
