"""bench.py prints ONE JSON line with the fields the benchmark contract names (GPU needed)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5",
                          "--cpu-baseline-seconds", "2", "--repeats", "5"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "cfg2" in d["config"]["workload"] and "model" not in d["config"]
    # value is whole-job throughput: images of all steps over the measured time
    assert abs(d["value"] - 64 * 40 / (d["ms_per_step"] * 40 * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["value_1core"] > 0
    t = d["timing"]
    assert t["repeats"] == 5 and t["ms_per_step_min"] <= d["ms_per_step"] <= t["ms_per_step_max"]
    assert d["config"]["input_sets_cycled"] * d["config"]["bytes_per_batch"] >= 512 << 20
    assert r["batches_per_launch"] == d["config"]["batches_per_launch"]
    # latency of one call beside the throughput: one batch and one image, microseconds
    lat = d["latency_us"]
    assert lat["one_batch"]["images"] == 64 and lat["one_image"]["images"] == 1
    assert 0 < lat["one_image"]["min"] <= lat["one_image"]["median"] < 1e4
    assert lat["one_image"]["median"] <= lat["one_batch"]["median"] * 1.5
    assert d["config"]["launches_in_flight"] == 2 and d["roofline"]["kernel_ms_alone"] > 0
