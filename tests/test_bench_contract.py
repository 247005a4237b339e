"""bench.py prints ONE JSON line carrying the driver's contract keys plus roofline / cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.check_output(
        [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
         "--seconds", "2", "--cpu-seconds", "1", "--no-other-configs"], text=True, cwd=REPO)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "acoustic_model" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # SURVEY section 8d's three end-points travel in the same line; `value` stays device-complete
    e = d["endpoints"]
    assert e["device_complete"]["value"] == d["value"]
    for k in ("from_pinned_host", "host_complete"):
        assert e[k]["unit"] == "frames/s" and e[k]["value"] > 0
    assert "time-capped" in c["sample"] and "time-capped" in d["cpu_baseline_all_cores"]["sample"]
    # value = frames of all steps / wall time
    assert abs(d["value"] - d["config"]["frames_per_gpu_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
