"""bench.py prints exactly one JSON line with the driver's contract keys (plus roofline / cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

gpu = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@gpu
def test_default_line_contract():
    d = run("--steps", "6", "--warmup", "2", "--no-cpu-baseline")
    assert REQUIRED <= set(d) and d["metric"] == "query_frames_per_sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 100 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] == "mfma" and 0.2 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_cpu_baseline_legs():
    """The cpu_baseline objects of the JSON line (host-only: the stock-torch restatements timed on this machine's cores)."""
    sys.path.insert(0, ROOT)
    import bench
    for cb, unit in ((bench.cpu_baseline(max_seconds=0.5), "frames/s"), (bench.spp_cpu_baseline(max_seconds=0.5), "images/s")):
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
        assert cb["unit"] == unit and cb["cores"] >= 1


@gpu
def test_extractor_and_pnp_lines():
    d = run("--extractor", "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert d["metric"] == "extractor_images_per_sec" and d["value"] > 100 and d["roofline"]["bound"] == "mfma"
    d = run("--pnp", "--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    assert d["metric"] == "pnp_solves_per_sec" and d["config"]["rotation_error_deg"] < 0.5
