"""GPU: bench.py prints exactly one JSON line with the fields the round driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_fields(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "512", "--passes", "3", "--steps", "2",
                          "--warmup", "1", "--cpu-samples", "40", "--cpu-threads", "2", "--stream-segments", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # value is cells per second of the timed region
    cells = d["config"]["windows_per_step"] * d["config"]["samples"]
    assert abs(d["value"] - cells * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    # secondary legs are reported beside the headline, never instead of it
    assert d["contexts_per_gpu"] == 1
    cs = d["concurrent_segments"]
    assert cs["contexts_per_gpu"] == 3 and cs["value"] > 0 and cs["unit"] == d["unit"]
