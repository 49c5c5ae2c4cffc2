"""Hand-derived known-answer tests that pin the oracle (the reference ships no golden vectors).

Each case is small enough to be computed by hand from the shader text (file:line cited); expected
values are closed-form, not produced by the code under test.
"""
import math

import numpy as np
import pytest

SH_C0 = 0.28209479177387814


def record(pos, log_scale=(-3.0, -3.0, -3.0), rot=(1, 0, 0, 0), opacity_logit=0.0, dc=(0, 0, 0), rest=None):
    r = np.zeros(62, np.float32)
    r[0:3] = pos
    r[6:9] = dc
    if rest is not None:
        r[9:54] = rest
    r[54] = opacity_logit
    r[55:58] = log_scale
    r[58:62] = rot
    return r


def frame(oracle, recs, w, h, cam=None):
    verts = oracle.activate_records(np.stack(recs))
    u = oracle.camera_uniforms(cam if cam is not None else oracle.default_camera(), w, h)
    return verts, u, oracle.stages(verts, u)


def test_default_camera_uniforms(oracle):
    """Renderer.cpp:719-754 with the default camera (Renderer.h:79-85): closed-form matrices."""
    w, h = 1920, 1080
    u = oracle.camera_uniforms(oracle.default_camera(), w, h)[0]
    tan_fovx = math.tan(math.radians(45.0) / 2)
    tan_fovy = tan_fovx * h / w
    assert u["tan_fovx"] == pytest.approx(tan_fovx, rel=1e-6)
    assert u["tan_fovy"] == pytest.approx(tan_fovy, rel=1e-6)
    view = u["view_mat"].reshape(4, 4).T  # column-major -> math
    np.testing.assert_allclose(view, np.diag([1.0, -1.0, -1.0, 1.0]), atol=0)
    proj = u["proj_mat"].reshape(4, 4).T
    n, f = 0.1, 1000.0
    expect = np.zeros((4, 4))
    expect[0, 0] = 1 / ((w / h) * tan_fovy)
    expect[1, 1] = -1 / tan_fovy  # row 1 negated (:747-750)
    expect[2, 2] = -(f + n) / (f - n)
    expect[2, 3] = -2 * f * n / (f - n)
    expect[3, 2] = -1
    np.testing.assert_allclose(proj, expect, rtol=2e-6, atol=1e-7)
    assert list(u["camera_position"]) == [0, 0, 0, 1]
    assert (u["width"], u["height"]) == (w, h)


def test_translated_camera_view_matrix(oracle):
    cam = oracle.default_camera(position=(1.0, 2.0, 3.0))
    u = oracle.camera_uniforms(cam, 640, 480)[0]
    view = u["view_mat"].reshape(4, 4).T
    expect = np.diag([1.0, -1.0, -1.0, 1.0])
    expect[:3, 3] = [-1.0, 2.0, 3.0]
    np.testing.assert_allclose(view, expect, atol=1e-6)


def test_activation(oracle):
    """GSScene.cpp:44-55: exp(scale), sigmoid(opacity), normalize(rot), planar->interleaved SH."""
    rest = np.arange(45, dtype=np.float32) + 100  # f_rest_k = 100 + k: R 100..114, G 115..129, B 130..144
    r = record((1, 2, 3), log_scale=(0.0, math.log(2.0), math.log(0.5)), rot=(2, 0, 0, 0), opacity_logit=0.0,
               dc=(7, 8, 9), rest=rest)
    v = oracle.activate_records(r[None])[0]
    np.testing.assert_array_equal(v["position"], [1, 2, 3, 1])
    np.testing.assert_allclose(v["scale_opacity"], [1.0, 2.0, 0.5, 0.5], rtol=1e-6)
    np.testing.assert_allclose(v["rotation"], [1, 0, 0, 0], rtol=1e-6)
    sh = v["sh"].reshape(16, 3)
    np.testing.assert_array_equal(sh[0], [7, 8, 9])
    for j in range(1, 16):
        np.testing.assert_array_equal(sh[j], [100 + j - 1, 115 + j - 1, 130 + j - 1])


def test_cov3d_axis_aligned_and_rotated(oracle):
    """precomp_cov3d.comp:31-47: Sigma = R S^2 R^T; a 90-degree turn about z swaps x and y."""
    s = (math.log(1.0), math.log(2.0), math.log(3.0))
    c45 = math.cos(math.pi / 4)
    v = oracle.activate_records(np.stack([record((0, 0, -5), s), record((0, 0, -5), s, rot=(c45, 0, 0, c45))]))
    cov = oracle.cov3d(v)
    np.testing.assert_allclose(cov[0], [1, 0, 0, 4, 0, 9], atol=1e-6)
    np.testing.assert_allclose(cov[1], [4, 0, 0, 1, 0, 9], atol=2e-6)


def test_on_axis_isotropic_gaussian(oracle):
    """A single isotropic Gaussian on the optical axis (preprocess.comp:130-178)."""
    w = h = 256
    z, sigma = 4.0, 0.05
    verts, u, st = frame(oracle, [record((0, 0, -z), (math.log(sigma),) * 3, opacity_logit=0.0, dc=(1, 0, -1))], w, h)
    a = st["attr"][0]
    f = w / (2 * math.tan(math.radians(45) / 2))
    var = (sigma * f / z) ** 2 + 0.3
    assert a["depth"] == pytest.approx(z, rel=1e-6)
    np.testing.assert_allclose(a["uv"], [(w - 1) / 2, (h - 1) / 2], atol=1e-4)  # ndc2Pix :110-113
    np.testing.assert_allclose(a["conic_opacity"], [1 / var, 0, 1 / var, 0.5], rtol=2e-5, atol=1e-7)
    lam = var + math.sqrt(0.1)  # mid^2 - det = 0 -> max(0.1, 0) :149
    radius = math.ceil(3 * math.sqrt(lam))
    assert a["color_radii"][3] == radius
    x0 = int(((w - 1) / 2 - radius) / 16)
    x1 = int(((w - 1) / 2 + radius + 15) / 16)
    np.testing.assert_array_equal(a["aabb"], [x0, x0, x1, x1])
    assert st["tiles"][0] == (x1 - x0) ** 2
    # colour: SH_C0*dc + 0.5, only R clamped at 0 (:102-104)
    np.testing.assert_allclose(a["color_radii"][:3], [SH_C0 * 1 + 0.5, 0.5, SH_C0 * -1 + 0.5], rtol=1e-6)
    # keys: x outer, y inner; key = tile<<32 | bits(depth) (preprocess_sort.comp:47-55)
    tiles_x = w // 16
    exp_tiles = [x + y * tiles_x for x in range(x0, x1) for y in range(x0, x1)]
    np.testing.assert_array_equal(st["keys"] >> np.uint64(32), exp_tiles)
    assert (st["keys"] & np.uint64(0xFFFFFFFF) == np.float32(z).view(np.uint32)).all()
    assert (st["payload"] == 0).all()


def test_pixel_centre_colour_and_alpha_cap(oracle):
    """Gaussian whose centre falls on an integer pixel: C = min(.99, o) * rgb at that pixel (render.comp:66-88)."""
    w = h = 256
    z = 4.0
    tan = math.tan(math.radians(45) / 2)
    k = 100  # target pixel column; uv = ((ndc+1)*W - 1)/2 = k  ->  ndc = (2k+1)/W - 1
    ndc = (2 * k + 1) / w - 1
    x = ndc * tan * z
    y = -ndc * tan * z  # view space is y-down (Renderer.cpp:738-745)
    for logit, alpha in ((0.0, 0.5), (30.0, 0.99)):
        verts, u, st = frame(oracle, [record((x, y, -z), (math.log(0.02),) * 3, opacity_logit=logit, dc=(1, 0.5, -0.5))], w, h)
        np.testing.assert_allclose(st["attr"][0]["uv"], [k, k], atol=2e-4)
        rgb = np.array([SH_C0 * 1 + 0.5, SH_C0 * 0.5 + 0.5, SH_C0 * -0.5 + 0.5])
        np.testing.assert_allclose(st["image"][k, k, :3], alpha * rgb, rtol=1e-4)
        assert st["image"][k, k, 3] == 1.0
        assert not st["image"][0, 0, :3].any()  # black background, alpha channel 1 (:98)
        assert (st["image"][..., 3] == 1.0).all()


def test_depth_cull_constant_and_offscreen(oracle):
    """p_view.z <= 0.2 culls (preprocess.comp:135), not the near plane; off-screen boxes have 0 tiles."""
    recs = [record((0, 0, -0.2)), record((0, 0, -0.25)), record((0, 0, 1.0)), record((50.0, 0, -4.0)),
            record((0, 0, -4.0))]
    verts, u, st = frame(oracle, recs, 128, 128)
    np.testing.assert_array_equal(st["tiles"] > 0, [False, True, False, False, True])


def test_front_to_back_order_and_tie_break(oracle):
    """Two opaque splats on the same pixel: nearer first; equal depth -> lower index first (stable sort)."""
    w = h = 64
    s = (math.log(0.3),) * 3
    red, blue = (3.0, -2.0, -2.0), (-2.0, -2.0, 3.0)
    # far (index 0, blue), near (index 1, red)
    _, _, st = frame(oracle, [record((0, 0, -5), s, opacity_logit=30, dc=blue), record((0, 0, -3), s, opacity_logit=30, dc=red)], w, h)
    c = st["image"][32, 32, :3]
    rgb_red = np.array([max(SH_C0 * 3 + 0.5, 0), SH_C0 * -2 + 0.5, SH_C0 * -2 + 0.5])
    rgb_blue = np.array([max(SH_C0 * -2 + 0.5, 0), SH_C0 * -2 + 0.5, SH_C0 * 3 + 0.5])
    a = 0.99 * math.exp(0)  # centre ~ half a pixel away; use the oracle's own alpha via ordering check instead
    assert c[0] > c[2]  # red dominates: it is blended first
    tile_of_centre = (32 // 16) + (32 // 16) * (w // 16)
    lo, hi = st["boundaries"][2 * tile_of_centre: 2 * tile_of_centre + 2]
    np.testing.assert_array_equal(st["sorted_payload"][lo:hi], [1, 0])
    # same depth: index order
    _, _, st2 = frame(oracle, [record((0, 0, -4), s, opacity_logit=30, dc=blue), record((0, 0, -4), s, opacity_logit=30, dc=red)], w, h)
    lo, hi = st2["boundaries"][2 * tile_of_centre: 2 * tile_of_centre + 2]
    np.testing.assert_array_equal(st2["sorted_payload"][lo:hi], [0, 1])
    assert st2["image"][32, 32, 2] > st2["image"][32, 32, 0]
    del a, rgb_red, rgb_blue


def test_transmittance_break_before_accumulate(oracle):
    """alpha = .99 twice: T' = 0.01*0.01 < 1e-4 in fp32 -> the second splat is dropped entirely (render.comp:82-88)."""
    w = h = 32
    s = (math.log(2.0),) * 3  # huge: alpha saturates at 0.99 over the whole frame
    recs = [record((0, 0, -3.0 - i), s, opacity_logit=40, dc=(1, 1, 1)) for i in range(3)]
    _, _, st = frame(oracle, recs, w, h)
    rgb = SH_C0 * 1 + 0.5
    t1 = np.float32(1.0) * (np.float32(1) - np.float32(0.99))
    assert t1 * (np.float32(1) - np.float32(0.99)) < np.float32(0.0001)  # second test_T trips the break
    np.testing.assert_allclose(st["image"][16, 16, :3], 0.99 * rgb, rtol=1e-5)


def test_full_screen_splat_tile_count(oracle):
    """A splat wider than the frame covers every tile: clamp to [0, tiles] (preprocess.comp:160-165)."""
    w, h = 640, 360
    _, _, st = frame(oracle, [record((0, 0, -3), (math.log(5.0),) * 3)], w, h)
    tx, ty = (w + 15) // 16, (h + 15) // 16
    assert st["tiles"][0] == tx * ty > 900
    np.testing.assert_array_equal(st["attr"][0]["aabb"], [0, 0, tx, ty])
    b = st["boundaries"].reshape(-1, 2)
    np.testing.assert_array_equal(b[:, 1] - b[:, 0], 1)  # one entry per tile


def test_scan_sort_ranges_small(oracle):
    tiles = np.array([0, 3, 0, 2, 5], np.uint32)
    np.testing.assert_array_equal(oracle.inclusive_scan(tiles), [0, 3, 3, 5, 10])
    keys = np.array([(2 << 32) | 7, (1 << 32) | 9, (2 << 32) | 7, (0 << 32) | 1, (1 << 32) | 3], np.uint64)
    pay = np.arange(5, dtype=np.uint32)
    sk, sp = oracle.sort_pairs(keys, pay)
    np.testing.assert_array_equal(sp, [3, 4, 1, 0, 2])  # stable: equal keys keep input order
    np.testing.assert_array_equal(sk, np.sort(keys, kind="stable"))
    b = oracle.tile_boundary(sk, 4)
    np.testing.assert_array_equal(b.reshape(4, 2), [[0, 1], [1, 3], [3, 5], [0, 0]])  # absent tile -> (0,0)
    assert not oracle.tile_boundary(np.zeros(0, np.uint64), 3).any()


def test_exp_definition(oracle):
    """gso_exp (the fast reading's polynomial): exact at 0, < 2.5 ULP on the range the blend uses, inside GLSL's
    3+2|x| ULP everywhere.  (The default reading's exp is gso_expf_libm: tests/test_expf_libm.py.)"""
    assert oracle.exp(np.float32(0.0)) == 1.0
    x = np.linspace(-8, 0, 20001).astype(np.float32)
    e = oracle.exp(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    ulp = np.abs(e - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() < 2.5
    x = np.linspace(-87, 0, 5001).astype(np.float32)
    e = oracle.exp(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    ulp = np.abs(e - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert (ulp <= 3 + 2 * np.abs(x)).all()
    assert oracle.exp(np.float32(-1000.0)) < 1e-37  # clamped, no wrap-around


def test_bgra8_pack(oracle):
    rgba = np.array([[[0.0, 0.5, 1.0, 1.0], [-0.2, 1.7, 0.25, 1.0]]], np.float32)
    out = oracle.pack_bgra8(rgba)
    np.testing.assert_array_equal(out[0, 0], [255, 128, 0, 255])
    np.testing.assert_array_equal(out[0, 1], [64, 255, 0, 255])


def test_simd_blend_is_bit_identical_to_the_scalar_checker(pkg, oracle):
    """gso_render_simd (AVX2, the CPU baseline's blend) must equal gso_render (the parity checker) bit for bit,
    also on ragged sizes and with non-finite inputs."""
    import __graft_entry__ as entry
    synth = entry.load_package().synth
    for n, w, h, seed, poison in [(6000, 256, 256, 0, False), (2500, 333, 177, 5, True), (9000, 130, 61, 9, False)]:
        rec = synth.synth_records(n, seed=seed, kind="A")
        if poison:
            rec[::50, 54] = np.nan
            rec[::77, 0] = np.inf
            rec[::91, 55] = np.nan
        verts = oracle.activate_records(rec)
        st = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))
        for contract, exp_mode in [(False, 2), (True, 0), (False, 0), (True, 2)]:  # the default reading first
            with oracle.reading(contract, exp_mode):
                a = oracle.render(st["attr"], st["boundaries"], st["sorted_payload"], w, h)
                b = oracle.render(st["attr"], st["boundaries"], st["sorted_payload"], w, h, simd=True)
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
            if (contract, exp_mode) == (False, 2):
                np.testing.assert_array_equal(a.view(np.uint32), st["image"].view(np.uint32))
