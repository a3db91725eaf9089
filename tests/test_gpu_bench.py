"""bench.py contract on the GPU: one JSON line with the driver's fields plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--shapes-per-step", "1"], ["--workload", "nearest"]])
def test_bench_prints_one_json_line_with_the_contract_fields(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--ddnm-steps", "2", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "shapes/hour" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 3600e3 * d["config"]["shapes_per_step"] / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    if "nearest" not in extra:
        assert r["bound"] == "mfma" and r["achieved"] > 100 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert d["config"]["shapes_per_step"] == (1 if "--shapes-per-step" in extra else 4)
