"""Generate the committed golden fixture from the oracle (run in the CPU container):

    python tests/golden/make_golden.py

The reference ships no golden data, so this pins *our* oracle (and through the GPU tests the HIP path)
against regressions: scene records, camera, uniforms, per-stage integer outputs and the fp32 image.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
oracle = entry.load_oracle()

W, H = 176, 112
rec = pkg.synth.synth_records(1200, seed=2024, kind="A")
q = np.array([0.97, -0.08, 0.15, 0.05], np.float32)
q /= np.linalg.norm(q)
cam = oracle.default_camera(position=(0.15, 0.1, 0.3), rotation=tuple(q))
verts = oracle.activate_records(rec)
u = oracle.camera_uniforms(cam, W, H)
st = oracle.stages(verts, u)
np.savez_compressed(os.path.join(HERE, "scene_a1200.npz"), records=rec, camera=cam, uniforms=u, tiles=st["tiles"],
                    sorted_tile=(st["sorted_keys"] >> np.uint64(32)).astype(np.uint32),
                    sorted_payload=st["sorted_payload"], boundaries=st["boundaries"],
                    image=st["image"][..., :3].astype(np.float32))
print("V", int((st["tiles"] > 0).sum()), "D", len(st["keys"]), "bytes", os.path.getsize(os.path.join(HERE, "scene_a1200.npz")))
