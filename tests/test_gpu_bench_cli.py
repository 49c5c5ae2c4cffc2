"""bench.py's command-line paths that the default run does not take, on small workloads (the default run is the driver's)."""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=500):
    e = dict(os.environ)
    e.pop("GS_SORT_PATH", None)
    e.pop("GS_EXP_MODE", None)  # bench.py sets the blend's modes itself
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_takes_a_ply(pkg, gpu, tmp_path, _sort_path):
    """BASELINE configs[2] is a trained scene's PLY (none ships with the reference or this container): `--ply FILE` and
    GS_SCENE=FILE bench whatever file is given -- GSScene::load on rank 0, the same file through the checker's own reader for the
    CPU baseline and the parity block."""
    if _sort_path == "1":
        pytest.skip("independent of the depth-order path: runs once")
    n = 30_000
    ply = str(tmp_path / "scene.ply")
    pkg.synth.write_ply(ply, pkg.synth.synth_records(n, seed=7, kind="T"))
    common = ["--width", "640", "--height", "360", "--steps", "20", "--warmup", "5"]
    for args, env in ((["--ply", ply], None), ([], {"GS_SCENE": ply})):
        b = _bench(args + common, env={"GS_CPU_BASELINE_SECONDS": "1", **(env or {})})
        assert b["config"]["scene"] == "ply" and b["config"]["gaussians"] == n and "scene.ply" in b["config"]["workload"]
        assert "configs[2]" in b["config"]["workload"] and "other_configs" not in b
        helpers.assert_value_is_frames_over_time(b)
        p = b["parity"]
        assert p["default"]["max_abs_vs_reference_text"] <= 1e-5 and p["default"]["pixels_above_1e-5"] == 0
        assert p["exact"]["bit_identical"] is True
        assert b["cpu_baseline"]["value"] > 0 and f"N={n}" in b["cpu_baseline"]["sample"]
        assert b["roofline"]["bound"] in ("hbm", "valu")  # (no committed pair count for this workload: the HBM view)


def test_bench_modes_and_other_configs_switch(pkg, gpu, _sort_path):
    """--exact benches the bit-identical blend (its own frames/s becomes `value`, the default mode's appears beside it);
    `other_configs` belongs to the headline workload only."""
    if _sort_path == "1":
        pytest.skip("independent of the depth-order path: runs once")
    b = _bench(["--gaussians", "40000", "--width", "640", "--height", "360", "--steps", "20", "--warmup", "5", "--exact", "--no-cpu-baseline"])
    assert "exp mode 2" in b["config"]["blend"] and b["frames_per_s_exact"] is None and b["frames_per_s_default"] > 0
    assert "other_configs" not in b and "cpu_baseline" not in b and "parity" not in b
    assert b["roofline"]["kernel"] in ("k_blend<2, false, false>", "k_preprocess") and b["timed"]["outliers"] is not None
