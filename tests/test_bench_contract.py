"""The bench line's contract (the driver parses it): keys and types of the `roofline` object, built here from the
committed counter summaries without a GPU, and -- on the GPU box -- of the whole line of a short default run."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name,ms,alg", [("current_pmc.json", 0.0766, 597120000), ("pmc_10k.json", 0.70, 5905536000)])
def test_roofline_object_keeps_the_contract_keys(name, ms, alg):
    b = _bench()
    pm = json.load(open(os.path.join(ROOT, "profiles", name)))
    r = b.roofline_object("k_mix_levels", {"alg_bytes": alg * 20, "launches": 20, "ms": ms * 20}, 20, 2 * ms, alg, 1, pm, 70000)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - alg / (ms * 1e-3) / 1e9) < 1.0
    if r["achieved"] <= r["peak"]:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    else:  # the contract figure exceeds the peak: no fraction, the reason instead
        assert r["frac"] is None and r["algorithmic_over_peak"] > 1.0 and "frac_note" in r
    assert isinstance(r["traffic"], int) and r["traffic"] > 0
    v = r["valu"]
    assert 0.0 < v["valu_busy"] <= 1.0 and v["calibration_probe"] == "mixlike" and r["limiter"] in ("valu", "hbm")


def test_no_committed_kernel_reads_more_than_fully_busy():
    """the calibrated VALU reading of every kernel of every committed counter summary is a
    fraction, and every probe of the calibration run reads 0.94-0.98 raw (1.00 +- 0.03 after the division by the
    mix-matched probe)."""
    cal = json.load(open(os.path.join(ROOT, "profiles", "valu_calibration.json")))["probes"]
    assert len(cal) == 13
    for name, p in cal.items():
        assert 0.93 <= p["valu_busy_raw"] <= 0.99, (name, p)
    for f in ("current_pmc.json", "pmc_10k.json", "pmc_flat.json", "pmc_config4.json"):
        pm = json.load(open(os.path.join(ROOT, "profiles", f)))
        for k, e in pm["kernels"].items():
            if "valu" in e:
                assert 0.0 <= e["valu"].get("valu_busy", e["valu"]["valu_busy_raw"]) <= 1.0, (f, k)


@pytest.mark.gpu
def test_default_bench_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--reps", "3", "--no-side"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(d["value"] - d["steps"] * 74496000 / (d["ms_per_step"] * d["steps"] * 1e-3) / 1e6) / d["value"] < 0.01
    assert "qt_adapter" in d["through_abi"] and "u8_dc_ms_per_frame" in d["through_abi"]
