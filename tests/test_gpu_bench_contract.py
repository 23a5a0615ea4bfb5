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


def run_bench(*args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True,
                       text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_on_a_small_scene():
    j = run_bench("--tess", "16", "--rays-log2", "18", "--steps", "3", "--warmup", "1", "--no-other-configs")
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


def test_default_line_carries_every_configuration():
    """The driver's line (here on a reduced scene) embeds the rest of the metric: any-hit on the same batch and
    configs[1], [4], [0], each with a rate, a kernel time, a cpu_baseline and a parity object without mismatches."""
    j = run_bench("--tess", "16", "--rays-log2", "18", "--steps", "2", "--warmup", "1")
    oc = j["config"]["other_configs"]
    assert set(oc) >= {"any_hit_same_batch", "configs[1]", "configs[4]", "configs[0]"}
    for key in ("any_hit_same_batch", "configs[1]", "configs[4]", "configs[0]"):
        e = oc[key]
        assert e["value"] > 0 and e["ms_per_step"] > 0 and e["roofline"] and e["cpu_baseline"]["value"] > 0, key
        for k, v in e["parity"].items():
            if "mismatches" in k:
                assert v == 0, (key, k, v)
        if "max_rel_err" in e["parity"]:
            assert e["parity"]["max_rel_err"] <= 1e-5
    a = oc["any_hit_same_batch"]
    assert a["parity"]["rays_checked_against_closest_hit_records"] == 1 << 18 and a["roofline"]["kernel_ms"] > 0
    assert oc["configs[1]"]["roofline"]["cache_resident"] is True


def test_gpus_2_starts_two_ranks_by_itself():
    """Plain `bench.py --gpus 2` (no launcher, WORLD_SIZE unset): two ranks on this box's one GPU over gloo, the
    sharded strong-scaling step with its gather, and the gathered records of both ranges equal to a single trace."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GIGL_BENCH_BACKEND"] = "gloo"
    j = run_bench("--gpus", "2", "--tess", "16", "--total-rays-log2", "18", "--steps", "2", "--warmup", "1", env=env)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["rays_total"] == 1 << 18
    assert j["config"]["rays_rank0"] == 1 << 17
    g = j["config"]["gather_check"]
    assert g["records_differing"] == 0 and g["rays_checked"] == 1 << 17
    assert j["parity"]["id_mismatches"] == 0
