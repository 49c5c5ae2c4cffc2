"""Shared helpers for the parity tests: oracle <-> device layout conversion."""
import numpy as np


def oracle_frame(oracle, records, width, height, camera=None):
    verts = oracle.activate_records(records)
    cam = camera if camera is not None else oracle.default_camera()
    u = oracle.camera_uniforms(cam, width, height)
    return verts, u, oracle.stages(verts, u)


def expected_depth_order(attr, tiles):
    """Visible Gaussian ids ascending by (bits(depth), id) -- what a stable sort of the reference's
    64-bit keys implies inside every tile."""
    vis = np.nonzero(tiles)[0].astype(np.uint32)
    bits = attr["depth"][vis].view(np.uint32)
    order = np.lexsort((vis, bits))
    return vis[order]


def compare_stages(pkg, rend, u, ref):
    """Bit-exact comparison of every stage tap with the oracle's buffers."""
    attr, tiles = ref["attr"], ref["tiles"]
    vis = tiles != 0
    np.testing.assert_array_equal(rend.stage("tiles"), tiles)
    np.testing.assert_array_equal(rend.stage("depth")[vis].view(np.uint32), attr["depth"][vis].view(np.uint32))
    np.testing.assert_array_equal(rend.stage("radius")[vis], attr["color_radii"][vis, 3])
    np.testing.assert_array_equal(rend.stage("aabb").reshape(-1, 4)[vis], attr["aabb"][vis].astype(np.uint16))
    np.testing.assert_array_equal(rend.stage("conic_opacity").reshape(-1, 4)[vis].view(np.uint32),
                                  attr["conic_opacity"][vis].view(np.uint32))
    uv_rg = rend.stage("uv_rg").reshape(-1, 4)[vis]
    np.testing.assert_array_equal(uv_rg[:, :2].view(np.uint32), np.ascontiguousarray(attr["uv"][vis]).view(np.uint32))
    np.testing.assert_array_equal(uv_rg[:, 2:].view(np.uint32),
                                  np.ascontiguousarray(attr["color_radii"][vis, :2]).view(np.uint32))
    np.testing.assert_array_equal(rend.stage("b")[vis].view(np.uint32),
                                  np.ascontiguousarray(attr["color_radii"][vis, 2]).view(np.uint32))
    st = rend.stats()
    if st.sort_path == 1:  # the global depth order exists as a buffer on that path only (gs_set_sort_path)
        order = expected_depth_order(attr, tiles)
        np.testing.assert_array_equal(rend.stage("depth_order"), order)
    assert st.num_visible == int(vis.sum())
    assert st.num_instances == len(ref["keys"])
    np.testing.assert_array_equal(rend.stage("sorted_tile"), (ref["sorted_keys"] >> np.uint64(32)).astype(np.uint32))
    np.testing.assert_array_equal(rend.stage("sorted_gid"), ref["sorted_payload"])
    np.testing.assert_array_equal(rend.stage("ranges", u), ref["boundaries"])
