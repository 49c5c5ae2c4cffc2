"""Timing on a clustered scene (uneven tile loads): 1 M Gaussians in 40 blobs, 1920x1080.  Diagnostic only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
n, w, h = 1_000_000, 1920, 1080
rec = pkg.synth.synth_records(n, seed=0, kind="S")
rng = np.random.default_rng(1)
centres = np.stack([rng.uniform(-3.5, 3.5, 40), rng.uniform(-2.0, 2.0, 40), rng.uniform(-11, -3, 40)], axis=1)
which = rng.integers(0, 40, n)
rec[:, 0:3] = centres[which] + rng.normal(0, float(sys.argv[1]) if len(sys.argv) > 1 else 0.25, (n, 3)).astype(np.float32)
scene = pkg.Scene.from_records(rec)
rend = pkg.Renderer(scene)
u = pkg.camera_uniforms(pkg.make_camera(), w, h)
for _ in range(3):
    rend.render_host(u)
st = rend.stats()
rg = rend.stage("ranges", u).reshape(-1, 2)
ln = rg[:, 1] - rg[:, 0]
print(f"V={st.num_visible} E1={st.num_bin_entries} D={st.num_instances} list len mean={ln.mean():.0f} max={ln.max()} "
      f"retries={st.retries}")
print({k: round(getattr(st, 'ms_' + k), 3) for k in ("preprocess", "prefix_sum", "preprocess_sort", "sort", "tile_boundary", "render", "total")})
