"""bench.py contract check on the GPU box: one JSON line with the driver's keys, `roofline` and `cpu_baseline`.

Runs the real script as a child process (short step count, bounded CPU sample) so a change in the engine's
outputs cannot break the round-end bench unnoticed.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2"] + extra
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_default_line():
    out = _run(["--cpu-seconds", "2"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["data"] == "synthetic"
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["peak"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = out["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    # the strict-f32 and the range-safe split mode ride in the same line, and so do the reference-facing calls
    for m in ("f32", "bf16x3"):
        assert out["modes"][m]["value"] > 0 and out["modes"][m]["roofline"]["frac"] > 0 and out["modes"][m]["laneconv_layer_us"]
    assert out["net_forward_dropin_ms"] > 0 and out["train_step_ms"] > 0, out.get("extra_timings_error")
    assert set(out["laneconv"]["layer_us"]) == {"fused", "tiled"}


def test_bench_single_stream_eager_mapnet():
    out = _run(["--streams", "1", "--no-graph", "--mapnet-only", "--workload", "S1", "--cpu-seconds", "0"])
    assert out["value"] > 0 and "S1" in out["config"]["workload"]
