"""The bench line the driver parses: checked on the committed artifact of the last GPU run (profiles/r01_bench.json) so that a change
of bench.py's output format is caught on the CPU.  (The numbers themselves are produced on the MI355X.)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _line("r01_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and "bs=1" in d["config"]["workload"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["n_gpus"]) < 0.02 * d["n_gpus"]          # value = images / time of the timed K steps
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 157.3
    assert r["traffic"] is None or r["traffic"] > 1e8
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["sequential"]["images_per_s"] <= d["value"] * 1.05


def test_bf16_line_is_labelled_as_such():
    d = _line("r01_bench_bf16.json")
    assert d["dtype"] == "bf16" and "bf16" in d["config"]["workload"] and d["roofline"]["peak"] == 2500.0
