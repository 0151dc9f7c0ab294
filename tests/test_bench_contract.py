"""bench.py prints ONE JSON line with the driver's contract keys plus the `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--width", "320", "--height", "180", "--cpu-threads", "4"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 4 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c
    assert d["parity"]["bit_exact_pixels"] == 1.0 and d["parity"]["rmse"] == 0.0


def test_bench_argparse_defaults():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert 'default=1)' in src.split('"--gpus"')[1][:60]
