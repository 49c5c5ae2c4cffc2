"""Committed golden fixture (tests/golden/scene_a1200.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_a1200.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def test_oracle_reproduces_golden(oracle, gold):
    verts = oracle.activate_records(gold["records"])
    u = oracle.camera_uniforms(gold["camera"], int(gold["uniforms"]["width"][0]), int(gold["uniforms"]["height"][0]))
    assert u.tobytes() == gold["uniforms"].tobytes()
    st = oracle.stages(verts, u)
    np.testing.assert_array_equal(st["tiles"], gold["tiles"])
    np.testing.assert_array_equal((st["sorted_keys"] >> np.uint64(32)).astype(np.uint32), gold["sorted_tile"])
    np.testing.assert_array_equal(st["sorted_payload"], gold["sorted_payload"])
    np.testing.assert_array_equal(st["boundaries"], gold["boundaries"])
    np.testing.assert_array_equal(st["image"][..., :3], gold["image"])


@pytest.mark.gpu
def test_hip_reproduces_golden(pkg, gpu, gold):
    w, h = int(gold["uniforms"]["width"][0]), int(gold["uniforms"]["height"][0])
    scene = pkg.Scene.from_records(gold["records"], device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(gold["camera"].view(pkg.binding.CAMERA_DT), w, h)
    assert u.tobytes() == gold["uniforms"].tobytes()
    img, _ = rend.render_host(u)
    np.testing.assert_array_equal(rend.stage("tiles"), gold["tiles"])
    np.testing.assert_array_equal(rend.stage("sorted_tile"), gold["sorted_tile"])
    np.testing.assert_array_equal(rend.stage("sorted_gid"), gold["sorted_payload"])
    np.testing.assert_array_equal(rend.stage("ranges", u), gold["boundaries"])
    assert np.abs(img[..., :3] - gold["image"]).max() <= 1e-4
