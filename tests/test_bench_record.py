"""The bench line committed with the profiles carries every field the measurement contract names."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_default.json"))[-1]
    with open(os.path.join(ROOT, "profiles", newest)) as fh:
        d = json.load(fh)
    assert d["ranks_seen"] == 1
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert d["config"]["workload"].startswith("cover")
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] * d["ms_per_step"] * 1e3 - d["config"]["segments_per_frame_per_gpu"]) < 0.01 * d["value"] * d["ms_per_step"] * 1e3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] == 1
