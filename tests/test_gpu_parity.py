"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, stage by stage and end to end.

Tolerances: integer / index stages bit-exact; preprocess floats bit-exact (same operation order,
no contraction); final fp32 RGB: max abs <= 1e-4 (BASELINE.json north_star), and in practice 0 -- the oracle's
default reading is bit-identical to the reference's shader text compiled for the CPU (tests/test_oracle_vs_ref.py), and
the product's default blend to both (the opt-in fast modes: tests/test_gpu_blend_modes.py).
"""
import os

import numpy as np
import pytest

from helpers import assert_guarded_close, assert_images_identical, compare_stages, oracle_frame

pytestmark = pytest.mark.gpu

PIXEL_TOL = 1e-4


def _run(pkg, oracle, rec, w, h, camera=None, default_blend_too=True):
    verts, u_ref, ref = oracle_frame(oracle, rec, w, h, camera)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    cam = camera if camera is not None else pkg.make_camera()
    u = pkg.camera_uniforms(cam, w, h)
    assert u.tobytes() == u_ref.tobytes(), "Renderer::updateUniforms restatements disagree"
    img, bgra = rend.render_host(u, want_rgba=True, want_bgra=True)
    # ... and the same frame in the library's DEFAULT blend (exp mode 3: the kernel that ships; the suite's renderers start in mode 2 for the
    # bitwise comparisons, conftest.py): rounding noise from the checker, no flip budget.  (Leaves the renderer in mode 2; the stage taps
    # of the callers read lists, which no blend mode touches.)
    if default_blend_too:
        assert_guarded_close(rend, u, ref["image"], label=f"{len(rec)} Gaussians @ {w}x{h}, default blend")
    return scene, rend, u, ref, img, bgra


def test_config_a_stages_and_pixels(pkg, oracle, gpu):
    """BASELINE configs[0]: 10 k Gaussians, 256x256, default camera."""
    rec = pkg.synth.synth_records(10000, seed=0, kind="A")
    scene, rend, u, ref, img, bgra = _run(pkg, oracle, rec, 256, 256)
    np.testing.assert_array_equal(scene.download_vertices().view(np.uint32),
                                  oracle.activate_records(rec).view(np.float32).reshape(-1, 60).view(np.uint32))
    np.testing.assert_array_equal(scene.download_cov3d().view(np.uint32), ref["cov3d"].view(np.uint32))
    compare_stages(pkg, rend, u, ref)
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    np.testing.assert_array_equal(bgra, oracle.pack_bgra8(ref["image"]))


def test_hip_against_the_reference_shader_text(pkg, oracle, gpu):
    """HIP path vs oracle/_ref (the reference's .comp files compiled for the CPU) with no oracle in between:
    config A and a rotated camera.  Integer stages, preprocess floats AND the image bit for bit."""
    import __graft_entry__ as entry
    refl = entry.load_ref()
    if not refl.available():
        pytest.skip("oracle/_ref did not travel to this box")
    q = np.array([0.9, 0.1, -0.3, 0.05], np.float32)
    q /= np.linalg.norm(q)
    for n, w, h, cam in [(10000, 256, 256, None), (8000, 640, 360, dict(position=(0.3, -0.2, 0.5), rotation=tuple(q)))]:
        rec = pkg.synth.synth_records(n, seed=n, kind="A")
        verts = oracle.activate_records(rec)
        ocam = oracle.default_camera(**cam) if cam else oracle.default_camera()
        u_ref = oracle.camera_uniforms(ocam, w, h)
        sr = refl.stages(verts, u_ref)
        scene = pkg.Scene.from_records(rec, device=0)
        rend = pkg.Renderer(scene)
        u = pkg.camera_uniforms(pkg.make_camera(**cam) if cam else pkg.make_camera(), w, h)
        assert u.tobytes() == u_ref.tobytes()
        img, _ = rend.render_host(u)
        np.testing.assert_array_equal(scene.download_cov3d().view(np.uint32), sr["cov3d"].view(np.uint32))
        compare_stages(pkg, rend, u, sr)
        assert_images_identical(img, sr["image"], label=f"HIP vs reference text {w}x{h}")
        print(f"HIP vs reference shader text, {n} @ {w}x{h}: every stage and the image bit-identical")


@pytest.mark.parametrize("w,h", [(200, 120), (33, 17), (16, 16), (1, 1), (641, 359), (4100, 2200), (7680, 4320),
                                 (16384, 48)])
def test_ragged_resolutions(pkg, oracle, gpu, w, h):
    """Odd sizes, single-tile frames, and every bin size of the tile binning (4x4, 8x8, 16x16, 32x32 tiles; both
    widths of the padded bin grid)."""
    rec = pkg.synth.synth_records(3000, seed=3, kind="A")
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, w, h)
    compare_stages(pkg, rend, u, ref)
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL


@pytest.mark.parametrize("shift", [2, 3, 4, 5])
def test_every_bin_size_on_one_frame(pkg, oracle, gpu, monkeypatch, shift):
    """GS_BIN_SHIFT forces bins of 4, 8, 16 and 32 tiles on the same 1080p frame (k_bin_build<1>, <1>, <4>, <16>;
    30 x 17 down to 4 x 3 bins): identical lists, ranges and pixels."""
    monkeypatch.setenv("GS_BIN_SHIFT", str(shift))
    rec = pkg.synth.synth_records(30000, seed=17, kind="A")
    rec[:40, 55:58] = 0.5  # a few splats that cover many bins
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 1920, 1080)
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))


def test_rotated_translated_camera(pkg, oracle, gpu):
    rec = pkg.synth.synth_records(5000, seed=5, kind="A")
    q = np.array([0.9, 0.1, -0.3, 0.05], np.float32)
    q /= np.linalg.norm(q)
    cam = pkg.make_camera(position=(0.3, -0.2, 0.5), rotation=tuple(q))
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 320, 200, camera=cam)
    compare_stages(pkg, rend, u, ref)
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL


@pytest.mark.parametrize("dense_min", ["0", "1000000000"])
def test_scene_read_in_spatial_order_changes_nothing(pkg, oracle, gpu, monkeypatch, dense_min):
    """Scenes of >= GS_SPATIAL_MIN Gaussians (default 4 M) are read by the per-frame kernels from a second copy in Morton order
    (a wave's 64 Gaussians neighbours in space); ids, taps, downloads and the broadcast blob keep the scene's own order.  Forced
    here on small scenes, with the dense lists on and off: every stage tap, the image, the equal-depth tie order, the vertex and
    cov3D downloads and the blob are what they are without it -- and what the oracle says."""
    monkeypatch.setenv("GS_L1_DENSE_MIN", dense_min)
    rec = pkg.synth.synth_records(60000, seed=23, kind="T")
    rec[100:140, 2] = rec[100, 2]          # a run of equal depths in front of the default camera: ties go by the scene's ids
    rec[100:140, 0:2] = rec[100, 0:2] + np.linspace(0, 0.02, 40)[:, None].astype(np.float32)
    w, h = 800, 450
    monkeypatch.setenv("GS_SPATIAL_MIN", "1000000000")
    plain = pkg.Scene.from_records(rec, device=0)
    monkeypatch.setenv("GS_SPATIAL_MIN", "0")
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, w, h)
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    np.testing.assert_array_equal(rend.stage("alpha_cut")[ref["tiles"] != 0].view(np.uint32),
                                  oracle.alpha_cut(oracle.activate_records(rec)["scale_opacity"][:, 3])[ref["tiles"] != 0].view(np.uint32))
    np.testing.assert_array_equal(scene.download_vertices().view(np.uint32), plain.download_vertices().view(np.uint32))
    np.testing.assert_array_equal(scene.download_cov3d().view(np.uint32), plain.download_cov3d().view(np.uint32))
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    blobs = []
    for s_ in (scene, plain):
        ptr, floats = s_.blob()
        host = np.empty(floats, np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(floats * 4), ctypes.c_int(2)) == 0
        blobs.append(host)
    np.testing.assert_array_equal(blobs[0].view(np.uint32), blobs[1].view(np.uint32))  # what a broadcast replicates: the scene's order
    rend.set_frames_in_flight(3)
    for _ in range(5):
        again, _ = rend.render_host(u)
    np.testing.assert_array_equal(again.view(np.uint32), ref["image"].view(np.uint32))
    scene.quantize_sh()  # the binary16 SH block follows the order the kernels read in
    q, _ = rend.render_host(u)
    plain.quantize_sh()
    r2 = pkg.Renderer(plain)
    r2.set_exp_mode(2)
    q2, _ = r2.render_host(u)
    np.testing.assert_array_equal(q.view(np.uint32), q2.view(np.uint32))


@pytest.mark.parametrize("dense_min", ["0", "1000000000"])
def test_level1_over_the_dense_lists_and_over_the_planes(pkg, oracle, gpu, monkeypatch, dense_min):
    """Level 1 of the bin-local path takes its items from the dense lists of visible Gaussians k_preprocess appends to
    (scenes of >= GS_L1_DENSE_MIN Gaussians; default 4 M, so the small scenes of this file would never use them) or from the
    N-wide planes.  Both forced here on the same scenes: stages and pixels equal the oracle's; the lists' counters are
    zeroed by each frame's last kernel, so frame after frame on the same buffers -- serial, four in flight, after an
    overflow re-run, after a resolution change -- must stay equal."""
    monkeypatch.setenv("GS_L1_DENSE_MIN", dense_min)
    rec = pkg.synth.synth_records(10000, seed=0, kind="A")
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 256, 256)
    compare_stages(pkg, rend, u, ref)
    for _ in range(3):
        again, _ = rend.render_host(u)
        np.testing.assert_array_equal(again.view(np.uint32), ref["image"].view(np.uint32))
    assert rend.stats().num_visible == int((ref["tiles"] != 0).sum())
    # many bins, a few splats that cover many of them (the wave-cooperative emission), 300 level-1 blocks' worth of lists
    rec = pkg.synth.synth_records(300000, seed=17, kind="A")
    rec[:40, 55:58] = 0.5
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 1920, 1080)
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    # other poses on the same renderer, four frames in flight, then a smaller target
    w, h = 1920, 1080
    poses = [pkg.camera_uniforms(pkg.make_camera(rotation=pkg.dist.pose_quaternion(k, 3.0)), w, h) for k in range(6)]
    serial = [rend.render_host(p)[0] for p in poses]
    rend.set_frames_in_flight(4)
    dev = _HipBuffers()
    outs = [dev.alloc(w * h * 16) for _ in poses]
    for p, o in zip(poses, outs):
        rend.render(p, o, 0)
    rend.synchronize()
    for s_img, o in zip(serial, outs):
        np.testing.assert_array_equal(dev.download(o, (h, w, 4), np.float32), s_img)
    dev.close()
    ref3 = oracle.stages(oracle.activate_records(rec), poses[3].view(oracle.UNIFORMS_DT))["image"]
    np.testing.assert_array_equal(serial[3].view(np.uint32), ref3.view(np.uint32))
    u_small = pkg.camera_uniforms(pkg.make_camera(), 640, 360)
    small, _ = rend.render_host(u_small)
    ref_small = oracle.stages(oracle.activate_records(rec), u_small.view(oracle.UNIFORMS_DT))["image"]
    np.testing.assert_array_equal(small.view(np.uint32), ref_small.view(np.uint32))


def test_empty_and_all_culled(pkg, oracle, gpu):
    # every Gaussian behind the camera: D = 0, black frame (SURVEY §8a edge cases)
    rec = pkg.synth.synth_records(500, seed=2, kind="A")
    rec[:, 2] = np.abs(rec[:, 2])
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 64, 48)
    st = rend.stats()
    assert st.num_visible == 0 and st.num_instances == 0
    assert not img[..., :3].any() and (img[..., 3] == 1).all()
    np.testing.assert_array_equal(rend.stage("ranges", u), np.zeros_like(ref["boundaries"]))
    # n = 0 scene
    scene0 = pkg.Scene.from_records(np.zeros((0, 62), np.float32))
    r0 = pkg.Renderer(scene0)
    img0, _ = r0.render_host(u)
    assert not img0[..., :3].any()


def test_instance_overflow_regrows(pkg, oracle, gpu):
    """A few huge splats: D >> initial capacity -> grow + re-run (Renderer.cpp:541-563)."""
    rec = pkg.synth.synth_records(300, seed=7, kind="A")
    rec[:, 55:58] = 1.5  # log-scale: sigma ~ 4.5 world units -> every splat covers the screen
    w, h = 1920, 1080
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, w, h)
    st = rend.stats()
    assert st.num_instances == len(ref["keys"])
    assert st.num_instances > (1 << 20) and st.retries >= 1
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL
    np.testing.assert_array_equal(rend.stage("sorted_gid"), ref["sorted_payload"])


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4097])
def test_small_counts_around_wave_and_tile_boundaries(pkg, oracle, gpu, n):
    """Visible counts around the wave (64) and radix-tile (2048) sizes, where ranking / chunking code changes regime."""
    rec = pkg.synth.synth_records(n, seed=100 + n, kind="A")
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 160, 96)
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))


def test_degenerate_keys(pkg, oracle, gpu):
    """All Gaussians at one depth (every radix digit constant, ties broken by id) and coincident Gaussians."""
    rec = pkg.synth.synth_records(5000, seed=8, kind="A")
    rec[:, 2] = -4.0          # identical view depth for the default camera
    rec[2500:, 0:3] = rec[:2500, 0:3]  # pairs of coincident centres
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 256, 144)
    assert len(np.unique(rend.stage("depth")[rend.stage("tiles") > 0])) == 1
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))


@pytest.mark.parametrize("layout", ["two_walls", "thin_slab", "wall_and_fog", "short_ties"])
def test_depth_distributions_against_the_bucket_order(pkg, oracle, gpu, layout):
    """k_bin_fast orders a bin by cutting its depth range into 4096 buckets and ranking inside the buckets; when a bucket is
    crowded it falls back to stable passes.  Depth distributions chosen to hit every branch: two thin walls far apart (two
    buckets hold everything: the fallback), a slab thinner than 4096 key values (the bucket index is the key offset itself),
    a wall inside a fog (one crowded bucket among sparse ones), and many short runs of exactly equal depths (the tie step
    after either order).  Lists, ranges and image must equal the oracle's, i.e. the reference's stable sort."""
    n = 12000
    rec = pkg.synth.synth_records(n, seed=77, kind="A")
    rng = np.random.default_rng(77)
    if layout == "two_walls":
        rec[:, 2] = np.where(rng.random(n) < 0.5, -2.5, -9.0) + rng.uniform(-2e-4, 2e-4, n)
    elif layout == "thin_slab":
        base = np.float32(-4.0).view(np.uint32)
        rec[:, 2] = (base + rng.integers(0, 900, n).astype(np.uint32)).view(np.float32)  # 900 adjacent binary32 values
    elif layout == "wall_and_fog":
        rec[: n // 2, 2] = -5.0 + rng.uniform(-1e-4, 1e-4, n // 2)
    else:
        rec[:, 2] = -np.round(rng.uniform(2.2, 9.0, n) * 400) / 400  # ~2700 distinct depths: runs of 4-5 equal keys
    rec[:, 0:2] *= 0.6  # denser bins
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 512, 288)
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))


def test_one_dense_bin(pkg, oracle, gpu):
    """Thousands of splats inside one 128x128-pixel bin: long chunk lists in a single bin, empty bins elsewhere."""
    rec = pkg.synth.synth_records(20000, seed=9, kind="A")
    rec[:, 0] = rec[:, 0] * 0.02 + 0.3
    rec[:, 1] = rec[:, 1] * 0.02 - 0.2
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, 1280, 720)
    compare_stages(pkg, rend, u, ref)
    assert rend.stats().num_bin_entries >= rend.stats().num_visible
    assert np.abs(img - ref["image"]).max() <= PIXEL_TOL


class _HipBuffers:
    """Device allocations through the HIP runtime the library itself uses (no torch: its bundled runtime and
    the system one cannot both own the GPU in one process when the library initialises first)."""

    def __init__(self):
        import ctypes
        self.C = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.ptrs = []

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), self.C.c_size_t(nbytes)) == 0
        self.ptrs.append(p)
        return p.value

    def download(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype)
        assert self.hip.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), self.C.c_void_p(ptr), self.C.c_size_t(out.nbytes), 2) == 0
        return out

    def close(self):
        for p in self.ptrs:
            self.hip.hipFree(p)


def test_frames_in_flight_match_serial(pkg, oracle, gpu):
    """Eight queued frames on four streams (different poses, distinct targets) == the same poses one at a time."""
    rec = pkg.synth.synth_records(30000, seed=10, kind="A")
    w, h = 640, 360
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    poses = [pkg.camera_uniforms(pkg.make_camera(rotation=pkg.dist.pose_quaternion(k, 2.0)), w, h) for k in range(8)]
    serial = [rend.render_host(u)[0] for u in poses]
    rend.set_frames_in_flight(4)
    dev = _HipBuffers()
    outs = [dev.alloc(w * h * 16) for _ in poses]
    for u, o in zip(poses, outs):
        rend.render(u, o, 0)
    rend.synchronize()
    for s_img, o in zip(serial, outs):
        np.testing.assert_array_equal(dev.download(o, (h, w, 4), np.float32), s_img)
    dev.close()
    ref = oracle.stages(oracle.activate_records(rec), poses[3].view(oracle.UNIFORMS_DT))["image"]
    assert np.abs(serial[3] - ref).max() <= PIXEL_TOL


def test_non_finite_inputs_do_not_break_parity(pkg, oracle, gpu):
    """NaN / Inf / huge values in the scene: nothing hangs, and every stage still equals the oracle bit for bit
    (such Gaussians end up culled or contribute garbage identically on both sides)."""
    rec = pkg.synth.synth_records(4000, seed=14, kind="A")
    rec[10, 0] = np.nan            # position
    rec[11, 2] = np.inf
    rec[12, 55:58] = 40.0          # exp(40): enormous splat
    rec[13, 55:58] = -60.0         # exp(-60): degenerate covariance
    rec[14, 58:62] = 0.0           # zero quaternion -> normalisation divides by zero
    rec[15, 54] = np.nan           # opacity
    rec[16, 6:9] = 1e30            # colour
    rec[17, 2] = -0.2000001        # just beyond the 0.2 depth cull
    verts = oracle.activate_records(rec)
    u = oracle.camera_uniforms(oracle.default_camera(), 320, 192)
    ref = oracle.stages(verts, u)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    img, _ = rend.render_host(pkg.camera_uniforms(pkg.make_camera(), 320, 192))
    np.testing.assert_array_equal(rend.stage("tiles"), ref["tiles"])
    np.testing.assert_array_equal(rend.stage("sorted_gid"), ref["sorted_payload"])
    np.testing.assert_array_equal(rend.stage("ranges", u), ref["boundaries"])
    both_nan = np.isnan(img) & np.isnan(ref["image"])
    np.testing.assert_array_equal(np.where(both_nan, 0, img).view(np.uint32), np.where(both_nan, 0, ref["image"]).view(np.uint32))


def test_renderer_reuse_across_resolutions(pkg, oracle, gpu):
    """One renderer, changing framebuffer sizes and frames in flight: buffers are re-sized behind queued frames."""
    rec = pkg.synth.synth_records(8000, seed=15, kind="A")
    verts = oracle.activate_records(rec)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(3)
    dev = _HipBuffers()
    sizes = [(320, 200), (1920, 1080), (64, 64), (4096, 2304), (320, 200), (800, 608)]
    targets = [(dev.alloc(w * h * 16), w, h) for w, h in sizes]
    for ptr, w, h in targets:
        rend.render(pkg.camera_uniforms(pkg.make_camera(), w, h), ptr, 0)
    rend.synchronize()
    for ptr, w, h in targets:
        ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))["image"]
        np.testing.assert_array_equal(dev.download(ptr, (h, w, 4), np.float32), ref)
    dev.close()


def test_overflow_with_frames_in_flight(pkg, oracle, gpu):
    """Four queued frames that all overflow the initial list capacity: every one is re-run after the buffers
    grow (the redo path of the frame ring), and each lands in its own target."""
    rec = pkg.synth.synth_records(300, seed=7, kind="A")
    rec[:, 55:58] = 1.5  # every splat covers the screen: D = 300 x 8160 > the initial capacity of 2^20
    w, h = 1920, 1080
    verts = oracle.activate_records(rec)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(4)
    dev = _HipBuffers()
    poses = [pkg.make_camera(position=(0.05 * k, 0.0, 0.0)) for k in range(4)]
    targets = [dev.alloc(w * h * 16) for _ in poses]
    for cam, ptr in zip(poses, targets):
        rend.render(pkg.camera_uniforms(cam, w, h), ptr, 0)
    rend.synchronize()
    st = rend.stats()
    assert st.retries >= 1 and st.num_instances > (1 << 20) and st.instance_capacity >= st.num_instances
    for k, ptr in enumerate(targets):
        ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(position=(0.05 * k, 0.0, 0.0)), w, h))["image"]
        np.testing.assert_array_equal(dev.download(ptr, (h, w, 4), np.float32), ref)
    dev.close()


def test_streamed_ply_ingest_matches_the_oracle_loader(pkg, oracle, gpu, tmp_path):
    """gs_scene_load_ply maps the file and streams it to HBM in 2^18-Gaussian chunks (GSScene::load's role,
    GSScene.cpp:26-68): the activated vertices in HBM must equal the oracle's loader bit for bit, across a chunk
    boundary, for the reference's layout and for a name-mapped one."""
    n = (1 << 18) + 12345
    rec = pkg.synth.synth_records(n, seed=41, kind="A")
    p_std = str(tmp_path / "std.ply")
    pkg.synth.write_ply(p_std, rec)
    want = oracle.load_ply(p_std).view(np.float32).reshape(-1, 60)
    sc = pkg.Scene.load_ply(p_std)
    assert sc.num_vertices == n
    np.testing.assert_array_equal(sc.download_vertices().view(np.float32).reshape(-1, 60), want)
    sc_rec = pkg.Scene.from_records(rec)
    np.testing.assert_array_equal(sc.download_cov3d(), sc_rec.download_cov3d())
    sc.close()
    sc_rec.close()
    # name-mapped: reversed property order, no normals, one extra double
    names = [nm for nm in pkg.synth._PROPS if nm not in ("nx", "ny", "nz")][::-1]
    dt = np.dtype([(nm, "<f4") for nm in names[:10]] + [("extra", "<f8")] + [(nm, "<f4") for nm in names[10:]])
    data = np.zeros(n, dt)
    for k, nm in enumerate(pkg.synth._PROPS):
        if nm in dt.names:
            data[nm] = rec[:, k]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join("property %s %s\n" % ("double" if nm == "extra" else "float", nm) for nm in dt.names) + "end_header\n"
    p_map = str(tmp_path / "mapped.ply")
    with open(p_map, "wb") as f:
        f.write(header.encode())
        f.write(data.tobytes())
    rec0 = rec.copy()
    rec0[:, 3:6] = 0  # normals absent
    want = oracle.activate_records(rec0).view(np.float32).reshape(-1, 60)
    sc = pkg.Scene.load_ply(p_map)
    np.testing.assert_array_equal(sc.download_vertices().view(np.float32).reshape(-1, 60), want)
    sc.close()
    # a truncated payload is an IO error, as in GSScene.cpp:36-41
    with open(p_std, "r+b") as f:
        f.truncate(os.path.getsize(p_std) - 100)
    with pytest.raises(pkg.GsError) as e:
        pkg.Scene.load_ply(p_std)
    assert e.value.code == -2


def test_blend_tuner_looks_again_on_the_device(pkg, gpu):
    """After 4096 settled frames the renderer measures the blend's two schedules again (gs_blend_tuner.h, kPeriod;
    tests/test_blend_tuner.py has the logic on the CPU): the settled flag drops and comes back, and the frame before,
    during and after the second look is the same frame."""
    rec = pkg.synth.synth_records(20000, seed=5, kind="T")
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    rend.set_exp_mode(3)
    rend.set_blend_lockstep(-1)
    rend.set_frames_in_flight(3)
    w, h = 320, 240
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    first = rend.render_host(u)[0]
    hb = _HipBuffers()
    ptrs = [hb.alloc(w * h * 16) for _ in range(3)]

    def run(n):
        for i in range(n):
            rend.render(u, ptrs[i % 3])
        rend.synchronize()

    run(320)  # (settled: 40 frames of hold, then one pass of ~70 frames, or two if lockstep wins the first)
    assert rend.blend_lockstep()[1] is True
    settled, during = [], None
    for _ in range(222):  # 4440 frames, looked at every 20 (a measurement lasts 70 to 140)
        run(20)
        settled.append(rend.blend_lockstep()[1])
        if not settled[-1] and during is None:
            during = hb.download(ptrs[19 % 3], (h, w, 4), np.float32)
    assert False in settled and settled[-1] is True
    assert 1 + 320 + 20 * (settled.index(False) + 1) >= 4096  # frames rendered when the second look was first seen: not before the period is over
    np.testing.assert_array_equal(during, first)
    np.testing.assert_array_equal(rend.render_host(u)[0], first)
    hb.close()
    rend.close()
    scene.close()


def test_spans_are_the_kernels_own_stamps(pkg, gpu, _sort_path):
    """Round 6: the six spans of gs_frame_stats are stamped by the kernels themselves (gs_kernels.h: FrameStamp) -- a pass's span runs
    from its first kernel's start to the next pass's, so the spans tile the frame exactly; on the global path the depth order's
    launches are part of `sort`; gs_set_timing(0) leaves the total; and the
    frame's total lies inside the wall clock around it (the stamps count a constant-rate clock: hipDeviceAttributeWallClockRate)."""
    import time
    rec = pkg.synth.synth_records(30000, seed=12, kind="A")
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    w, h = 640, 360
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    hb = _HipBuffers()
    ptr = hb.alloc(w * h * 16)
    for _ in range(5):
        rend.render(u, ptr)
    rend.synchronize()
    t0 = time.perf_counter()
    rend.render(u, ptr)
    rend.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    st = rend.stats()
    spans = [st.ms_preprocess, st.ms_prefix_sum, st.ms_preprocess_sort, st.ms_sort, st.ms_render]
    assert all(x > 0 for x in spans) and st.ms_tile_boundary == 0
    assert abs(sum(spans) - st.ms_total) <= 1e-3 * st.ms_total + 2e-5      # they tile the frame (float32 ms of 10-ns ticks)
    assert 0 < st.ms_total <= wall_ms * 1.05 + 0.05
    assert st.sort_path == (1 if _sort_path == "1" else 2)
    rend.set_timing(False)
    rend.render(u, ptr)
    rend.synchronize()
    st0 = rend.stats()
    assert st0.ms_total > 0 and st0.ms_render == 0 and st0.ms_preprocess == 0
    rend.set_timing(True)
    rend.render(u, ptr)
    rend.synchronize()
    assert rend.stats().ms_render > 0
    hb.close()
    rend.close()
    scene.close()


def test_frame_intervals_track_completions(pkg, gpu):
    """gs_get_frame_intervals: one completion-to-completion interval per consecutive pair of retired frames -- never negative
    (0 for a frame that had already finished when its predecessor did: frames on different streams complete out of order and
    the library measures against the latest completion so far, gs_renderer.cpp retire_oldest) -- and consistent with the wall
    clock of the queued batch (a generous bound: GPU timestamps against the host's clock, on whatever box this runs)."""
    import time
    rec = pkg.synth.synth_records(20000, seed=3, kind="A")
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(3)
    u = pkg.camera_uniforms(pkg.make_camera(), 320, 240)
    hb = _HipBuffers()
    ptrs = [hb.alloc(320 * 240 * 16) for _ in range(3)]
    for i in range(6):
        rend.render(u, ptrs[i % 3])
    rend.synchronize()
    rend.frame_intervals(reset=True)
    t0 = time.perf_counter()
    for i in range(50):
        rend.render(u, ptrs[i % 3])
    rend.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    iv = rend.frame_intervals(reset=True)
    assert len(iv) == 49 and (iv >= 0).all() and iv.sum() > 0
    assert iv.sum() <= wall_ms * 1.25 + 1.0
    assert len(rend.frame_intervals()) == 0
    rend.close()
    scene.close()
    hb.close()


def _dense_bin_records(pkg, n=20000):
    rec = pkg.synth.synth_records(n, seed=9, kind="A")
    rec[:, 0] = rec[:, 0] * 0.02 + 0.3
    rec[:, 1] = rec[:, 1] * 0.02 - 0.2
    rec[:, 2] = -4.0 + 0.05 * rec[:, 2]  # a thin slab: every splat's tile box lands in the same bins
    return rec


def test_sort_paths_agree_and_fall_back(pkg, oracle, gpu, monkeypatch):
    """gs_set_sort_path: the bin-local path (one in-LDS order per bin) and the global depth order build the same
    lists.  A bin beyond 16384 candidates is taken in depth slabs (level 4, still bin-local) up to 65535; beyond that it is
    an error when the bin-local path is forced and a transparent re-run on the global path in automatic mode; once the
    bins fit again for 32 frames the automatic mode steps back."""
    monkeypatch.delenv("GS_SORT_PATH", raising=False)
    w, h = 640, 360
    # (1) a scene whose bins all fit: both forced paths, stage by stage
    rec = pkg.synth.synth_records(6000, seed=21, kind="A")
    verts, u_ref, ref = oracle_frame(oracle, rec, w, h)
    scene = pkg.Scene.from_records(rec)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    for mode in (2, 1, 0):
        rend = pkg.Renderer(scene)
        rend.set_sort_path(mode)
        img, _ = rend.render_host(u)
        assert rend.stats().sort_path == (1 if mode == 1 else 2)
        assert rend.stats().max_bin_entries <= 16384
        compare_stages(pkg, rend, u, ref)
        np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
        rend.close()
    scene.close()
    # (2) 20 000 and 40 000 splats in one bin: depth slabs, forced or automatic
    for n in (20000, 40000):
        rec = _dense_bin_records(pkg, n)
        verts, u_ref, ref = oracle_frame(oracle, rec, w, h)
        scene = pkg.Scene.from_records(rec)
        for mode in (2, 0):
            rend = pkg.Renderer(scene)
            rend.set_sort_path(mode)
            img, _ = rend.render_host(u)
            st = rend.stats()
            assert st.sort_path == 2 and st.sort_level == 4 and st.retries >= 1 and 16384 < st.max_bin_entries <= 65535, \
                (st.sort_path, st.sort_level, st.retries, st.max_bin_entries)
            compare_stages(pkg, rend, u, ref)
            np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
            rend.close()
        scene.close()
    # (3) 70 000 splats in one bin: beyond the slabs
    rec = _dense_bin_records(pkg, 70000)
    verts, u_ref, ref = oracle_frame(oracle, rec, w, h)
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    rend.set_sort_path(2)
    with pytest.raises(pkg.GsError) as e:
        rend.render_host(u)
    assert e.value.code == -5
    rend.close()
    rend = pkg.Renderer(scene)  # automatic
    img, _ = rend.render_host(u)
    st = rend.stats()
    assert st.sort_path == 1 and st.retries >= 1 and st.max_bin_entries > 65535
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    # (4) the camera turns away from the dense bin: after 32 fitting frames the bin-local path is back
    away = pkg.make_camera(rotation=(0.0, 0.0, 1.0, 0.0))  # looking down +z: nothing in view
    ua = pkg.camera_uniforms(away, w, h)
    for _ in range(40):
        rend.render_host(ua)
    assert rend.stats().sort_path == 2
    img2, _ = rend.render_host(u)  # back at the dense bin: falls back again, same image
    assert rend.stats().sort_path == 1
    np.testing.assert_array_equal(img2.view(np.uint32), ref["image"].view(np.uint32))
    rend.close()
    scene.close()


def test_depth_slabs_with_concentrated_depths(pkg, oracle, gpu, monkeypatch):
    """The slab cut is made on depth buckets: a dense bin whose candidates crowd into ONE bucket (a wall seen face on,
    30 000 splats within 1e-4 of one depth) cannot be cut and goes to the global path; the same bin with two such walls
    plus a spread-out rest is cut between them.  Both must give the oracle's lists and pixels."""
    monkeypatch.setenv("GS_SORT_PATH", "0")
    w, h = 640, 360
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    rng = np.random.default_rng(5)
    for name, depth in [("one wall", lambda n: -4.0 + rng.uniform(-5e-5, 5e-5, n)),
                        ("two walls and fog", lambda n: np.where(rng.random(n) < 0.3, -3.0, np.where(rng.random(n) < 0.5, -6.0, rng.uniform(-9, -2.2, n)))
                                                        + rng.uniform(-5e-5, 5e-5, n))]:
        rec = _dense_bin_records(pkg, 30000)
        rec[:, 2] = depth(len(rec))
        verts, u_ref, ref = oracle_frame(oracle, rec, w, h)
        scene = pkg.Scene.from_records(rec)
        rend = pkg.Renderer(scene)
        img, _ = rend.render_host(u)
        st = rend.stats()
        print(f"{name}: path {st.sort_path} level {st.sort_level} fullest bin {st.max_bin_entries} retries {st.retries}")
        assert st.max_bin_entries > 16384
        compare_stages(pkg, rend, u, ref)
        np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
        rend.close()
        scene.close()


def test_bin_local_sort_at_its_capacity(pkg, oracle, gpu, monkeypatch):
    """A bin with 9 000-16 000 candidates (more than one 8-round batch per wave) still takes the bin-local path."""
    monkeypatch.delenv("GS_SORT_PATH", raising=False)
    rec = _dense_bin_records(pkg, n=13000)
    w, h = 640, 360
    verts, u_ref, ref = oracle_frame(oracle, rec, w, h)
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    rend.set_sort_path(2)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    img, _ = rend.render_host(u)
    st = rend.stats()
    assert st.sort_path == 2 and 8192 < st.max_bin_entries <= 16384
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    rend.close()
    scene.close()


def test_many_path_fallbacks_do_not_exhaust_a_lifetime_budget(pkg, oracle, gpu, monkeypatch):
    """A fly-through that enters and leaves a dense bin: every entry re-runs the frame at the level the bin asks for (depth
    slabs: one redo each), every exit steps back down after 32 fitting frames.  More than 70 such round
    trips must neither raise GS_ERR_OVERFLOW (the guard counts CONSECUTIVE re-runs of a frame) nor change a pixel."""
    monkeypatch.setenv("GS_SORT_PATH", "0")  # automatic path choice is what is under test
    rec = pkg.synth.synth_records(30000, seed=21, kind="A")
    rec[:, 0] = rec[:, 0] * 0.01 + 0.2     # all of them inside one bin: 30 000 candidates > kBinSortMax
    rec[:, 1] = rec[:, 1] * 0.01 - 0.1
    w, h = 512, 288
    verts = oracle.activate_records(rec)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    dense = pkg.camera_uniforms(pkg.make_camera(), w, h)
    away = pkg.camera_uniforms(pkg.make_camera(rotation=pkg.dist.pose_quaternion(18, 10.0)), w, h)  # looking backwards
    dev = _HipBuffers()
    target = dev.alloc(w * h * 16)
    for trip in range(72):
        rend.render(dense, target, 0)
        for _ in range(33):
            rend.render(away, target, 0)
    rend.render(dense, target, 0)
    rend.synchronize()
    st = rend.stats()
    assert st.retries >= 72, st.retries
    assert st.sort_path == 2 and st.sort_level == 4 and st.max_bin_entries > 16384  # 30 000 in one bin: depth slabs
    ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))
    np.testing.assert_array_equal(dev.download(target, (h, w, 4), np.float32), ref["image"])
    np.testing.assert_array_equal(rend.stage("sorted_gid"), ref["sorted_payload"])
    dev.close()


def test_candidate_overflow_on_fresh_buffers(pkg, oracle, gpu, monkeypatch):
    """Level-1 candidates AND instances beyond a tiny initial capacity, first frame of a renderer (buffers straight
    from hipMalloc): the bin kernels run over the unwritten gap before the frame is re-run, which must stay in
    bounds; the re-run frames are exact.  Several grow steps in a row stay inside the consecutive-redo guard."""
    monkeypatch.setenv("GS_INITIAL_CAPACITY", "1024")
    rec = pkg.synth.synth_records(4000, seed=22, kind="A")
    rec[:200, 55:58] = 1.0   # 200 screen-filling splats: they touch every bin
    w, h = 1280, 720
    verts = oracle.activate_records(rec)
    for flights in (1, 3):
        scene = pkg.Scene.from_records(rec, device=0)
        rend = pkg.Renderer(scene)
        rend.set_frames_in_flight(flights)
        u = pkg.camera_uniforms(pkg.make_camera(), w, h)
        img, _ = rend.render_host(u)
        st = rend.stats()
        ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))
        assert st.retries >= 1 and st.num_instances == len(ref["keys"]) and st.num_bin_entries > 1024
        assert st.instance_capacity >= st.num_instances
        compare_stages(pkg, rend, u, ref)
        np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
        rend.close()
        scene.close()


def test_needle_gaussians(pkg, oracle, gpu):
    """Thin, long, randomly oriented splats (sigma ratio up to e^7): far from the centre the terms of `power` are
    ~1e5 and cancel to a few units, so the per-quadrant culling bound must carry a term-scaled slack.  Bit-exact
    image, and the lists as usual."""
    for seed, n, w, h in [(31, 6000, 640, 360), (32, 20000, 1920, 1080)]:
        rec = pkg.synth.synth_records(n, seed=seed, kind="A")
        rng = np.random.default_rng(seed)
        rec[:, 55] = rng.uniform(-1.5, 0.0, n)          # long axis: sigma 0.2 .. 1 world units
        rec[:, 56:58] = rng.uniform(-9.0, -6.0, (n, 2))  # thin axes: sub-pixel
        rec[:, 58:62] = rng.normal(size=(n, 4))          # random orientation
        rec[:, 54] = rng.uniform(0.0, 4.0, n)            # fairly opaque, so that tau is large
        scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, w, h)
        compare_stages(pkg, rend, u, ref)
        np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))


def test_fp16_sh_storage_is_the_pipeline_on_rounded_coefficients(pkg, oracle, gpu):
    """gs_scene_quantize_sh (opt-in): preprocess reads binary16 SH.  The frame must equal, bit for
    bit, the reference pipeline run on the coefficients rounded to binary16 (nearest even) -- the oracle is fed
    exactly those -- and differ from the fp32 frame by no more than the rounding suggests."""
    rec = pkg.synth.synth_records(20000, seed=61, kind="A")
    rec[:50, 9:54] *= 1e-5  # some coefficients in binary16's subnormal range
    w, h = 640, 360
    verts = oracle.activate_records(rec)
    u_ref = oracle.camera_uniforms(oracle.default_camera(), w, h)
    ref32 = oracle.stages(verts, u_ref)
    verts16 = verts.copy()
    verts16["sh"] = verts["sh"].astype(np.float16).astype(np.float32)
    ref16 = oracle.stages(verts16, u_ref)
    scene = pkg.Scene.from_records(rec, device=0)
    assert scene.sh_bits == 32
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    img32, _ = rend.render_host(u)
    np.testing.assert_array_equal(img32.view(np.uint32), ref32["image"].view(np.uint32))
    scene.quantize_sh()
    assert scene.sh_bits == 16
    img16, _ = rend.render_host(u)
    compare_stages(pkg, rend, u, ref16)
    np.testing.assert_array_equal(img16.view(np.uint32), ref16["image"].view(np.uint32))
    d = np.abs(img16 - img32).max()
    assert 0 < d < 5e-3, d  # binary16 keeps 11 bits of each coefficient
    rend.close()
    scene.close()


def test_ply_larger_than_4_gib(pkg, oracle, gpu, tmp_path):
    """A 4.4 GiB PLY (17.9 M records; sparse on disk): the reference's Buffer takes a uint32_t size (Buffer.h:15) and
    would truncate; here offsets are 64-bit end to end -- header, mmap, chunked streaming, the blob.  Records are
    planted before, across and after the 4 GiB mark and read back from HBM."""
    n = 17_900_000
    path = str(tmp_path / "big.ply")
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n +
              "".join("property float %s\n" % p for p in pkg.synth._PROPS) + "end_header\n").encode()
    rec = pkg.synth.synth_records(6, seed=71, kind="A")
    where = [0, 1, (1 << 32) // 248, (1 << 32) // 248 + 1, n - 2, n - 1]  # record 17 318 416 straddles byte 2^32
    with open(path, "wb") as f:
        f.write(header)
        f.truncate(len(header) + n * 248)
        for r, i in zip(rec, where):
            f.seek(len(header) + i * 248)
            f.write(r.astype("<f4").tobytes())
    assert os.path.getsize(path) > (1 << 32)
    try:
        scene = pkg.Scene.load_ply(path, device=0)
    except pkg.GsError as e:  # a box without 4.3 GiB to spare for the blob is not what is under test
        if e.code == -4:
            pytest.skip("not enough device memory for a 17.9 M-Gaussian scene")
        raise
    assert scene.num_vertices == n
    want = oracle.activate_records(rec).view(np.float32).reshape(-1, 60)
    for k, i in enumerate(where):
        got = scene.download_vertex_range(i, 1)[0]
        np.testing.assert_array_equal(got.view(np.uint32), want[k].view(np.uint32), err_msg=f"record {i}")
    # an untouched record (all zero bytes): exp(0) scales, sigmoid(0) opacity, zero position
    z = scene.download_vertex_range(12345678, 1)[0]
    assert z[4] == 1.0 and z[5] == 1.0 and z[6] == 1.0 and z[7] == 0.5 and not z[:3].any()
    scene.close()


def test_graph_replay_mode(pkg, oracle, gpu, monkeypatch):
    """gs_set_graph_mode: a frame's launches are captured once per configuration and replayed; the uniforms and the
    targets of each frame come from the parameter block.  A moving camera with three frames in flight, a resolution
    change, a capacity overflow (grow + re-capture) and the fp16-SH switch must all give the oracle's frames."""
    monkeypatch.setenv("GS_INITIAL_CAPACITY", "200000")
    rec = pkg.synth.synth_records(20000, seed=81, kind="A")
    verts = oracle.activate_records(rec)
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    rend.set_graph_mode(True)
    rend.set_frames_in_flight(3)
    dev = _HipBuffers()
    for (w, h) in [(640, 360), (800, 448)]:   # the second size overflows the 200 000-entry capacity
        poses = [dict(position=(0.03 * k, -0.02 * k, 0.05 * k), rotation=pkg.dist.pose_quaternion(k, 1.5)) for k in range(7)]
        targets = [dev.alloc(w * h * 16) for _ in poses]
        for cam, ptr in zip(poses, targets):
            rend.render(pkg.camera_uniforms(pkg.make_camera(**cam), w, h), ptr, 0)
        rend.synchronize()
        for cam, ptr in zip(poses, targets):
            ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(**cam), w, h))["image"]
            np.testing.assert_array_equal(dev.download(ptr, (h, w, 4), np.float32), ref)
    assert rend.stats().retries >= 1
    # stage taps and stats still work in this mode -- since round 6 the per-pass spans too: the kernels stamp the frame's timeline
    # themselves (gs_kernels.h: FrameStamp), a replayed frame like any other
    u = pkg.camera_uniforms(pkg.make_camera(), 800, 448)
    img, _ = rend.render_host(u)
    ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), 800, 448))
    compare_stages(pkg, rend, u, ref)
    st = rend.stats()
    assert st.ms_total > 0 and 0 < st.ms_render <= st.ms_total and 0 < st.ms_preprocess < st.ms_total
    assert abs((st.ms_preprocess + st.ms_prefix_sum + st.ms_preprocess_sort + st.ms_sort + st.ms_render) - st.ms_total) <= 1e-3 * st.ms_total + 1e-4
    scene.quantize_sh()  # changes what preprocess reads: the captured launches must follow
    verts16 = verts.copy()
    verts16["sh"] = verts["sh"].astype(np.float16).astype(np.float32)
    img16, _ = rend.render_host(u)
    np.testing.assert_array_equal(img16.view(np.uint32), oracle.stages(verts16, oracle.camera_uniforms(oracle.default_camera(), 800, 448))["image"].view(np.uint32))
    rend.set_graph_mode(False)
    img2, _ = rend.render_host(u)
    np.testing.assert_array_equal(img2, img16)
    dev.close()
    rend.close()
    scene.close()


def test_level_changes_with_frames_in_flight_stay_bin_local(pkg, gpu, monkeypatch):
    """A fresh renderer, three frames in flight, a cluster whose fullest bin holds > 65535 candidates at 8 x 8 tiles and between
    16384 and 65535 at 4 x 4: the first frame climbs level 0 -> smaller bins at level 3 -> depth slabs (level 4) while the next
    frames are being queued.  A frame queued behind it must run with the level of after that climb; round 4's T(6e6) profile
    showed one being queued with the level of before it (the wait for buffer allocation sat between the decision and the
    launch), failing at once and sending the renderer to the global path for 32 frames."""
    monkeypatch.setenv("GS_SORT_PATH", "0")
    rec = pkg.synth.synth_records(360000, seed=91, kind="A")
    rec[:, 0] = rec[:, 0] * 0.35 + 0.3      # the cluster of test_bins_are_refined_before_the_global_path, 4.5 x as dense
    rec[:, 1] = rec[:, 1] * 0.35 - 0.2
    rec[:, 2] = -4.0 + 0.05 * rec[:, 2]
    rec[:, 55:58] -= 1.5
    w, h = 1920, 1080
    scene = pkg.Scene.from_records(rec)
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(3)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    hb = _HipBuffers()
    ptrs = [hb.alloc(w * h * 16) for _ in range(3)]
    for i in range(9):
        rend.render(u, ptrs[i % 3])
    rend.synchronize()
    st = rend.stats()
    assert st.sort_path == 2 and st.sort_level == 4 and st.bin_tiles == 4 and 16384 < st.max_bin_entries <= 65535, \
        (st.sort_path, st.sort_level, st.bin_tiles, st.max_bin_entries, st.retries)
    assert st.retries >= 2, st.retries  # smaller bins, then slabs
    imgs = [hb.download(p, (h, w, 4), np.float32) for p in ptrs]
    host, _ = rend.render_host(u)
    for img in imgs:
        np.testing.assert_array_equal(img.view(np.uint32), host.view(np.uint32))
    rend.close()
    scene.close()
    hb.close()


def test_bins_are_refined_before_the_global_path(pkg, oracle, gpu, monkeypatch):
    """Automatic mode, a cluster that puts > 16384 candidates into one 8 x 8-tile bin but fits once the bins are
    4 x 4 tiles: the frame is re-run with the smaller bins and stays on the bin-local path (what keeps 6 M-Gaussian
    scenes there); the lists and the pixels are the oracle's."""
    monkeypatch.setenv("GS_SORT_PATH", "0")
    rec = pkg.synth.synth_records(80000, seed=91, kind="A")
    rec[:, 0] = rec[:, 0] * 0.35 + 0.3      # a cluster ~400 px across at 1080p, depth 4: ~19 k candidates in the fullest
    rec[:, 1] = rec[:, 1] * 0.35 - 0.2      # 128-px bin, ~9 k in the fullest 64-px one (counted with the oracle's boxes)
    rec[:, 2] = -4.0 + 0.05 * rec[:, 2]
    rec[:, 55:58] -= 1.5                     # small splats: a tile box rarely spans two 64-px bins
    w, h = 1920, 1080
    scene, rend, u, ref, img, _ = _run(pkg, oracle, rec, w, h, default_blend_too=False)  # (this test reads the FIRST frame's statistics)
    st = rend.stats()
    assert st.retries >= 1 and st.sort_path == 2 and st.bin_tiles == 4, (st.retries, st.sort_path, st.bin_tiles, st.max_bin_entries)
    assert st.max_bin_entries <= 16384
    compare_stages(pkg, rend, u, ref)
    np.testing.assert_array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    # the refinement jumps to the largest in-LDS order (level 3); the first clean frame tells what the smaller bins need and the
    # next frame runs there, not 32 frames per step later
    limits = [4096, 8192, 12288, 16384]
    fits = min(lv for lv in range(4) if lv == 3 or st.max_bin_entries <= limits[lv] * 7 // 8)
    img2, _ = rend.render_host(u)
    st2 = rend.stats()
    assert st.sort_level == 3 and st2.sort_level == fits and st2.retries == st.retries, (st.sort_level, st2.sort_level, fits, st.max_bin_entries)
    np.testing.assert_array_equal(img2.view(np.uint32), img.view(np.uint32))
    assert_guarded_close(rend, u, ref["image"], label="refined bins, default blend")
