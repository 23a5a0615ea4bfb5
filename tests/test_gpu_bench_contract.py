"""bench.py's one-line contract with the driver, on a reduced workload so that it runs in seconds: one JSON
line on stdout with the agreed keys, a roofline and a cpu_baseline object, and a parity object without
mismatches (the bench checks its own output against the oracle outside the timed region)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True,
                       text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_on_a_small_scene():
    j = run_bench("--tess", "16", "--rays-log2", "18", "--steps", "3", "--warmup", "1")
    for k in KEYS:
        assert k in j, k
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"]
    assert j["value"] > 0 and abs(j["value"] - (1 << 18) / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["kernel_ms"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["kernel"].startswith("trace_kernel<") and r["waves_per_simd"] >= 1
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "Mrays/s" and c["sample"]
    p = j["parity"]
    assert p["rays_checked"] == 1 << 18 and p["id_mismatches"] == 0
    assert p["max_rel_err_t"] == 0.0 and p["max_rel_err_u"] == 0.0 and p["max_rel_err_v"] == 0.0


@pytest.mark.parametrize("config,unit", [("vpl", "kVPLs/s"), ("cornell_shadow", "Mrays/s"), ("transmittance", None)])
def test_other_configurations_print_the_same_schema(config, unit):
    j = run_bench("--config", config, "--steps", "2", "--warmup", "1")
    for k in KEYS:
        assert k in j, k
    assert (unit is None or j["unit"] == unit) and j["value"] > 0 and j["roofline"] and j["cpu_baseline"]["value"] > 0
    p = j["parity"]
    for k, v in p.items():
        if k.endswith("mismatches"):
            assert v == 0, (k, v)
    if "max_rel_err" in p:
        assert p["max_rel_err"] <= p["tolerance"] == 1e-5
