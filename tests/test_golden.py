"""Committed golden fixture (tests/golden/scene_a1200.npz, made by tests/golden/make_golden.py).

Every output in the file was computed by the reference's own shader text compiled for the CPU (oracle/_ref); the
restated oracle and the HIP path must reproduce it: bit for bit in cov3D, the visible VertexAttribute records,
tiles_overlap, the sorted payload, the tile boundaries AND the image (render.comp's exp() is libm's there; the oracle and
the HIP blend restate that function in binary64 and evaluate :66,87 uncontracted, like the CPU compilation).
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_a1200.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def test_fixture_was_made_from_the_reference_shader_text(gold):
    src = str(gold["generator"])
    for name in ("precomp_cov3d.comp", "preprocess.comp", "prefix_sum.comp", "preprocess_sort.comp",
                 "tile_boundary.comp", "render.comp", "common.glsl"):
        assert f"src/shaders/{name} sha256=" in src


def test_oracle_reproduces_golden(oracle, gold):
    verts = oracle.activate_records(gold["records"])
    u = oracle.camera_uniforms(gold["camera"], int(gold["uniforms"]["width"][0]), int(gold["uniforms"]["height"][0]))
    assert u.tobytes() == gold["uniforms"].tobytes()
    st = oracle.stages(verts, u)
    assert st["cov3d"].tobytes() == gold["cov3d"].tobytes()
    np.testing.assert_array_equal(st["tiles"], gold["tiles"])
    vis = st["tiles"] > 0
    for f in ("conic_opacity", "color_radii", "aabb", "uv", "depth"):
        assert np.ascontiguousarray(st["attr"][f][vis]).tobytes() == \
            np.ascontiguousarray(gold["visible_attr"][f]).tobytes(), f
    np.testing.assert_array_equal((st["sorted_keys"] >> np.uint64(32)).astype(np.uint32), gold["sorted_tile"])
    np.testing.assert_array_equal(st["sorted_payload"], gold["sorted_payload"])
    np.testing.assert_array_equal(st["boundaries"], gold["boundaries"])
    np.testing.assert_array_equal(st["image"][..., :3].view(np.uint32), gold["image"].view(np.uint32))


@pytest.mark.gpu
def test_hip_reproduces_golden(pkg, gpu, gold):
    w, h = int(gold["uniforms"]["width"][0]), int(gold["uniforms"]["height"][0])
    scene = pkg.Scene.from_records(gold["records"], device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(gold["camera"].view(pkg.binding.CAMERA_DT), w, h)
    assert u.tobytes() == gold["uniforms"].tobytes()
    img, _ = rend.render_host(u)
    np.testing.assert_array_equal(rend.stage("tiles"), gold["tiles"])
    vis = gold["tiles"] > 0
    va = gold["visible_attr"]
    np.testing.assert_array_equal(rend.stage("conic_opacity").reshape(-1, 4)[vis].view(np.uint32),
                                  np.ascontiguousarray(va["conic_opacity"]).view(np.uint32))
    np.testing.assert_array_equal(rend.stage("depth")[vis].view(np.uint32), va["depth"].view(np.uint32))
    np.testing.assert_array_equal(rend.stage("uv_rg").reshape(-1, 4)[vis][:, :2].view(np.uint32),
                                  np.ascontiguousarray(va["uv"]).view(np.uint32))
    np.testing.assert_array_equal(rend.stage("sorted_tile"), gold["sorted_tile"])
    np.testing.assert_array_equal(rend.stage("sorted_gid"), gold["sorted_payload"])
    np.testing.assert_array_equal(rend.stage("ranges", u), gold["boundaries"])
    np.testing.assert_array_equal(np.ascontiguousarray(img[..., :3]).view(np.uint32), gold["image"].view(np.uint32))
