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
