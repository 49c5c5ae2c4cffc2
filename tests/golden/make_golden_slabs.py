"""Generate the second committed fixture -- a scene that forces DEPTH SLABS (depth-order level 4: k_bin_queue, or k_bin_slabs +
k_slab_work with GS_L2_QUEUE=0) -- from the REFERENCE'S OWN SHADER TEXT (run in the CPU container, where /root/reference is mounted):

    python tests/golden/make_golden_slabs.py

24 000 small splats (log-scales lowered by 1: the lists stay short, the fixture small) inside one bin of 4 x 4 tiles, their depths in two thin walls and a fog between them (the slab cut has to
fall between the walls).  The 6 MB of input records are NOT stored: they are regenerated from the committed generator
(`slab_scene_records` below, on 3dgs.cpp_amd/synth.py) and the fixture holds their SHA-256, so a test that feeds the kernels
something else than what the reference text was given fails on the hash.  Every OUTPUT in the file -- tiles_overlap, the visible
Gaussians' depths, the sorted payload (the per-tile lists in the reference's order), the tile boundaries and the fp32 image -- was
computed by oracle/_ref (src/shaders/*.comp compiled for the CPU, the eight radix passes as written), none by the oracle port.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

W, H, N = 640, 360, 24000


def slab_scene_records(pkg):
    """The fixture's input (deterministic: counter-based generator + a seeded numpy stream)."""
    rec = pkg.synth.synth_records(N, seed=9, kind="A")
    rec[:, 0] = rec[:, 0] * 0.02 + 0.3
    rec[:, 1] = rec[:, 1] * 0.02 - 0.2
    rec[:, 55:58] -= 1.0  # PLY log-scales
    rng = np.random.default_rng(5)
    pick, half = rng.random(N), rng.random(N)
    depth = np.where(pick < 0.3, -3.0, np.where(half < 0.5, -6.0, rng.uniform(-9, -2.2, N))) + rng.uniform(-5e-5, 5e-5, N)
    rec[:, 2] = depth.astype(np.float32)
    return np.ascontiguousarray(rec)


if __name__ == "__main__":
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    ref = entry.load_ref()
    rec = slab_scene_records(pkg)
    verts = oracle.activate_records(rec)
    cam = oracle.default_camera()
    u = oracle.camera_uniforms(cam, W, H)
    st = ref.stages(verts, u)
    assert st.get("text_sort"), "the radix passes must be the reference's text for this fixture"
    vis = st["tiles"] > 0
    assert len(rec) < 65536
    out = os.path.join(HERE, "scene_slabs24k.npz")
    np.savez_compressed(out, records_sha256=np.array(hashlib.sha256(rec.tobytes()).hexdigest()), camera=cam, uniforms=u,
                        tiles=st["tiles"].astype(np.uint16), visible_depth=st["attr"]["depth"][vis],
                        sorted_tile=(st["sorted_keys"] >> np.uint64(32)).astype(np.uint16),
                        sorted_payload=st["sorted_payload"].astype(np.uint16), boundaries=st["boundaries"],
                        image=st["image"][..., :3].astype(np.float32), generator=np.array(ref.sources()))
    print("V", int(vis.sum()), "D", len(st["keys"]), "bytes", os.path.getsize(out))
