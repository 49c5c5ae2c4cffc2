"""The bench line's contract (task statement), checked on the committed round-1 line and on bench.py's
own byte model -- no GPU needed."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("gs_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_bench_line_has_every_contract_field():
    with open(os.path.join(ROOT, "profiles", "r01_bench_default.json")) as f:
        b = json.load(f)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert base["metric"].startswith(b["metric"])  # the headline clause of BASELINE.json's metric
    assert b["unit"] == "frames/s" and b["higher_is_better"] is True
    assert b["n_gpus"] == 1 and b["scaling"] == "weak" and b["vs_baseline"] is None and b["data"] == "synthetic"
    assert b["dtype"] == "f32" and "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 1e3 / b["ms_per_step"]) / b["value"] < 1e-3  # whole-job frames / wall time
    r = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = b["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == b["unit"]
    # every reported number carries its measured N, V, D
    for key in ("gaussians", "visible", "instances", "tiles"):
        assert b["config"][key] > 0
    assert set(b["passes"]) == {"preprocess", "prefix_sum", "preprocess_sort", "sort", "tile_boundary", "render"}


def test_algorithmic_bytes_model():
    m = _bench_module()
    n, v, d, e1, t, p = 1_000_000, 483_640, 4_893_565, 741_254, 8160, 1920 * 1080
    g = m.algorithmic_bytes(n, v, d, e1, t, p, bin_local=False)
    b = m.algorithmic_bytes(n, v, d, e1, t, p, bin_local=True)
    assert g["render"] == b["render"] == 40 * d + 16 * p
    assert g["tile_boundary"] == b["tile_boundary"] == 12 * e1 + 12 * t
    assert g["preprocess"] == n * 40 + v * 248 and b["preprocess"] == g["preprocess"] + 4 * n
    assert g["sort"] - b["sort"] == 80 * v - 12 * e1  # four global passes over V against one in-LDS order of E1
    assert g["prefix_sum"] == 8 * v and b["prefix_sum"] == 8 * n
    assert all(x > 0 for x in list(g.values()) + list(b.values()))
