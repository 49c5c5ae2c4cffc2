"""A/B helper (GPU box): render config B with the library named by GS3D_HIP_LIB and compare with the oracle.
Prints max |d| away from render.comp's thresholds and the threshold-flip pixels (tests/helpers.compare_images)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from helpers import compare_images  # noqa: E402

pkg, oracle = entry.load_package(), entry.load_oracle()
n, w, h = int(os.environ.get("N", 1_000_000)), 1920, 1080
rec = pkg.synth.synth_records(n, seed=0, kind="S")
scene = pkg.Scene.from_records(rec, device=0)
rend = pkg.Renderer(scene)
u = pkg.camera_uniforms(pkg.make_camera(), w, h)
img, _ = rend.render_host(u)
ref = oracle.stages(oracle.activate_records(rec), oracle.camera_uniforms(oracle.default_camera(), w, h))
d = np.abs(img - ref["image"])
print("lib", os.environ.get("GS3D_HIP_LIB", "default"), "bit-equal", bool(np.array_equal(img.view(np.uint32), ref["image"].view(np.uint32))),
      "max", float(d.max()), "n>1e-6", int((d.max(axis=2) > 1e-6).sum()), "n>1e-4", int((d.max(axis=2) > 1e-4).sum()))
rest, flips = compare_images(img, ref["image"], ref, w, label="A/B")
print("  off-threshold max", rest, "flips", [(x, y, round(v, 6)) for x, y, v, _ in flips])
