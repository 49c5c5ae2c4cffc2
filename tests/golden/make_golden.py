"""Generate the committed golden fixture from the REFERENCE'S OWN SHADER TEXT (run in the CPU container, where
/root/reference is mounted):

    python tests/golden/make_golden.py

The reference ships no golden data and cannot be built here (no Vulkan / glslang / glm), so the vectors are made by
oracle/_ref: the reference's .comp files compiled for the CPU by oracle/build_ref.py (IEEE binary32, no contraction,
libm exp).  Inputs (PLY-domain records, camera) come from the synthetic generator; the camera uniforms come from the
restated glm arithmetic of the oracle (Renderer::updateUniforms needs glm, which is absent) and are stored too.
Every stage output in the file -- cov3D, the 64-byte VertexAttribute records of the visible Gaussians, tiles_overlap,
the sorted payload, the tile boundaries and the fp32 image -- is computed by the reference text, none by the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
oracle = entry.load_oracle()
ref = entry.load_ref()

W, H = 176, 112
rec = pkg.synth.synth_records(1200, seed=2024, kind="A")
q = np.array([0.97, -0.08, 0.15, 0.05], np.float32)
q /= np.linalg.norm(q)
cam = oracle.default_camera(position=(0.15, 0.1, 0.3), rotation=tuple(q))
verts = oracle.activate_records(rec)
u = oracle.camera_uniforms(cam, W, H)
st = ref.stages(verts, u)
vis = st["tiles"] > 0
out = os.path.join(HERE, "scene_a1200.npz")
np.savez_compressed(out, records=rec, camera=cam, uniforms=u, cov3d=st["cov3d"], tiles=st["tiles"],
                    visible_attr=st["attr"][vis],
                    sorted_tile=(st["sorted_keys"] >> np.uint64(32)).astype(np.uint32),
                    sorted_payload=st["sorted_payload"], boundaries=st["boundaries"],
                    image=st["image"][..., :3].astype(np.float32), generator=np.array(ref.sources()))
print("V", int(vis.sum()), "D", len(st["keys"]), "bytes", os.path.getsize(out))
print(ref.sources())
