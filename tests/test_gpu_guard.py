"""The library's default blend -- exp mode 3: the hardware's v_exp_f32 with the reference's decisions -- against the reference's
shader text, and the two mechanisms that make its decisions the reference's:

  * render.comp:78 (`alpha < 1/255 -> continue`) is decided on `power` against the Gaussian's ALPHA CUT, computed at load from the
    opacity with libm's expf (GS_STAGE_ALPHA_CUT): the tap must equal the checker's bisection (oracle.alpha_cut) and satisfy its
    defining property against THIS MACHINE's libm, edge opacities included;
  * render.comp:82 (`T (1 - alpha) < 1e-4 -> break`) is guarded: quadrants in which a pixel comes within the proven window of the
    threshold are re-rendered with exp mode 2's arithmetic.

The adversarial scene puts thousands of pixels on both cuts at once: a stack of screen-filling splats whose accumulated
transmittance crosses 1e-4 along dozens of rings (adjacent pixels a few 1e-4 apart in T), weak layers whose alpha crosses 1/255
along others, loud colours behind the break so that a single wrong decision moves a pixel by 1e-4 .. 1e-3.  The guarded blend
must stay within 1e-5 of the reference text there; the unguarded v_exp_f32 (mode 1) is shown not to.
"""
import ctypes

import numpy as np
import pytest

from helpers import GUARD_TOL, assert_guarded_close, assert_images_identical, compare_stages

pytestmark = pytest.mark.gpu
SH_C0 = 0.28209479177387814


def _vertices(pos, scale, opacity, rgb):
    """GSScene::Vertex records (60 floats): position(4) scale(3) opacity rotation(4) sh(48); view-independent colour rgb."""
    n = len(pos)
    v = np.zeros((n, 60), np.float32)
    v[:, 0:3] = pos
    v[:, 3] = 1.0
    v[:, 4:7] = np.asarray(scale, np.float32).reshape(n, -1)
    v[:, 7] = opacity
    v[:, 8] = 1.0
    v[:, 12:15] = (np.asarray(rgb, np.float64) - 0.5) / SH_C0  # SH degree 0: colour = SH_C0 * dc + 0.5
    return v


def adversarial_vertices(shift=(0.0, 0.0), strong=0.9003, weak_layers=60, seed=0):
    """One stack of screen-filling isotropic splats (sigma ~ 2500 px at 512 x 512).  Four strong layers: (1 - 0.9003 e^p)^4
    crosses 1e-4 on a ring of ~65 px radius; behind them `weak_layers` layers of opacity 0.00393 .. 0.0046 (alpha crosses 1/255 on
    rings of their own, and each kept layer moves T by 0.4 %: one more T-ring per layer) in colours of magnitude ~100."""
    rng = np.random.default_rng(seed)
    n = 4 + weak_layers
    z = -(5.0 + 0.01 * np.arange(n))
    pos = np.stack([np.full(n, shift[0]), np.full(n, shift[1]), z], axis=1)
    opacity = np.concatenate([np.full(4, strong), rng.uniform(0.00393, 0.0046, weak_layers)]).astype(np.float32)
    rgb = np.concatenate([rng.uniform(0.2, 1.0, (4, 3)), rng.uniform(-100.0, 100.0, (weak_layers, 3))])
    rgb[4:, 0] = np.abs(rgb[4:, 0])  # (the red channel is clamped at 0 by preprocess.comp)
    return _vertices(pos, np.full((n, 3), 20.0), opacity, rgb)


def _reference(oracle, verts, w, h):
    import __graft_entry__ as entry
    gsref = entry.load_ref()
    u = oracle.camera_uniforms(oracle.default_camera(), w, h)
    ref = oracle.stages(np.ascontiguousarray(verts).view(oracle.VERTEX_DT).reshape(-1), u)
    if gsref.available():
        rimg = gsref.render(ref["attr"], ref["boundaries"], ref["sorted_payload"], w, h)
        assert_images_identical(ref["image"], rimg, label="oracle vs render.comp on the adversarial scene")
    return u, ref


def _adversarial_case(pkg, oracle, case):
    w = h = 512
    rng = np.random.default_rng(100 + case)
    verts = adversarial_vertices(shift=tuple(rng.uniform(-0.3, 0.3, 2)), strong=float(rng.uniform(0.9001, 0.9012)), seed=case)
    u_ref, ref = _reference(oracle, verts, w, h)
    scene = pkg.Scene.from_vertices(verts, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    assert u.tobytes() == u_ref.tobytes()
    rend.set_exp_mode(2)
    exact, _ = rend.render_host(u)
    compare_stages(pkg, rend, u, ref)
    assert_images_identical(exact, ref["image"], label="exp mode 2 on the adversarial scene")
    # how adversarial: pixels whose break decision sits within 1e-6 / 1e-4 (relative) of the threshold, from a float64 trace
    near6, near4 = _pixels_near_the_t_cut(ref, w, h)
    assert near4 > 500, near4
    worst, redo, resolved = assert_guarded_close(rend, u, ref["image"], label=f"adversarial {case}: guarded blend")
    assert resolved + redo > 0  # the guard did act (what it resolved exactly is what keeps the frame within rounding noise)
    rend.set_exp_mode(1)  # the same exp WITHOUT the guard: the decisions it protects do flip here
    loose, _ = rend.render_host(u)
    d1 = np.abs(loose[..., :3].astype(np.float64) - ref["image"][..., :3]).max(axis=2)
    print(f"adversarial {case}: {near4} px within 1e-4 of the T cut, {near6} within 1e-6; guarded max|d| {worst:.3g}, {resolved} break decisions resolved exactly, "
          f"{redo} of {(w // 8) * (h // 8)} quadrants re-rendered; unguarded v_exp_f32: max|d| {d1.max():.3g}, {int((d1 > GUARD_TOL).sum())} px > 1e-5")
    rend.close()
    scene.close()
    return int((d1 > GUARD_TOL).sum())


def test_adversarial_scenes_on_both_cuts(pkg, oracle, gpu):
    """Six stacks (centre, strong opacity and weak layers vary): the guarded blend within 1e-5 of the reference text on each, and
    over the six the UNGUARDED hardware exp must get at least one break wrong (otherwise the scenes prove nothing)."""
    total = sum(_adversarial_case(pkg, oracle, case) for case in range(6))
    assert total >= 1, "no unguarded flip on any adversarial case: the scenes do not sit on the cut"


def _pixels_near_the_t_cut(ref, w, h):
    """float64 re-trace of the (single-stack) adversarial scene: per pixel the smallest |T (1 - alpha) / 1e-4 - 1| met."""
    attr = ref["attr"]
    order = ref["sorted_payload"][ref["boundaries"][0]:ref["boundaries"][1]]  # every tile holds the whole stack, in depth order
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    T = np.ones((h, w))
    alive = np.ones((h, w), bool)
    near = np.full((h, w), np.inf)
    for g in order:
        co = attr["conic_opacity"][g].astype(np.float64)
        dx, dy = attr["uv"][g][0] - xs, attr["uv"][g][1] - ys
        power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
        alpha = np.minimum(0.99, co[3] * np.exp(np.minimum(power, 0)))
        kept = alive & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        near = np.where(kept, np.minimum(near, np.abs(test_T / 1e-4 - 1)), near)
        brk = kept & (test_T < 1e-4)
        T = np.where(kept & ~brk, test_T, T)
        alive &= ~brk
    return int((near < 1e-6).sum()), int((near < 1e-4).sum())


def test_alpha_cut_tap_equals_the_checkers_and_libm(pkg, oracle, gpu):
    """GS_STAGE_ALPHA_CUT of every visible Gaussian == oracle.alpha_cut(opacity) (bisection over gso_expf_libm), and the cut's
    defining property holds against THIS MACHINE's libm: kept at the cut, not kept one binary32 below it."""
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    rec = pkg.synth.synth_records(20000, seed=3, kind="A")
    verts = oracle.activate_records(rec)
    opacity = verts["scale_opacity"][:, 3]  # (a view: GSScene::Vertex keeps the opacity in scale_opacity.w)
    # edge opacities on the first Gaussians: 0, negative, exactly 1/255 and its neighbours, 1, > 1, huge, inf, NaN, denormal
    edge = np.array([0.0, -0.5, 1 / 255, np.nextafter(np.float32(1 / 255), np.float32(0)), np.nextafter(np.float32(1 / 255), np.float32(1)),
                     1.0, 0.99, 0.9899999, 1.5, 1e30, np.inf, np.nan, 1e-40, 0.0039, 0.00393], np.float32)
    opacity[:len(edge)] = edge
    w = h = 256
    scene = pkg.Scene.from_vertices(verts, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    rend.render_host(u)
    vis = rend.stage("tiles") != 0
    cut = rend.stage("alpha_cut")
    want = oracle.alpha_cut(opacity)
    np.testing.assert_array_equal(cut[vis].view(np.uint32), want[vis].view(np.uint32))
    assert vis[:len(edge)].sum() >= 8  # most of the edge cases are on screen

    def kept(o, p):
        a = np.float32(o) * np.float32(libm.expf(float(p)))
        a = np.float32(0.99) if np.isnan(a) else min(np.float32(0.99), a)
        return not a < np.float32(1.0 / 255.0)
    for i in np.nonzero(vis)[0][:3000]:
        o, c = opacity[i], cut[i]
        if np.isposinf(c):
            assert not kept(o, np.float32(-0.0)), (o, c)
        elif np.isneginf(c):
            assert kept(o, np.float32(-np.inf)) and kept(o, np.float32(-50.0)), (o, c)
        else:
            below = np.nextafter(np.float32(c), np.float32(-np.inf))
            assert c <= 0 and kept(o, c) and not kept(o, below), (o, c)
    # and the frame with those opacities: mode 2 bit-identical to the checker's reference reading, mode 3 within rounding noise
    ref = oracle.stages(verts, oracle.camera_uniforms(oracle.default_camera(), w, h))
    rend.set_exp_mode(2)
    img, _ = rend.render_host(u)
    finite = np.isfinite(ref["image"]).all(axis=2)
    np.testing.assert_array_equal(np.isfinite(img).all(axis=2), finite)
    assert_images_identical(np.where(finite[..., None], img, 0), np.where(finite[..., None], ref["image"], 0), label="edge opacities, exp mode 2")
    rend.set_exp_mode(3)
    g, _ = rend.render_host(u)
    d = np.abs(np.where(finite[..., None], g, 0).astype(np.float64) - np.where(finite[..., None], ref["image"], 0))
    assert d.max() <= GUARD_TOL, d.max()
    rend.close()
    scene.close()


def test_guarded_blend_writes_the_same_bgra8_image_as_its_float_image(pkg, oracle, gpu):
    """The B8G8R8A8_UNORM target (render.comp:98 + Swapchain.cpp:22-28) in the default mode: the packed image equals the packing of
    the same frame's float image, and is within one LSB of the reference's packed image (rounding noise can cross a rounding
    boundary of the UNORM conversion, nothing more), on ragged sizes too."""
    for seed, (w, h) in enumerate([(256, 256), (333, 217), (1, 1), (17, 640)]):
        rec = pkg.synth.synth_records(8000, seed=40 + seed, kind="A")
        scene = pkg.Scene.from_records(rec, device=0)
        rend = pkg.Renderer(scene)
        rend.set_exp_mode(3)
        u = pkg.camera_uniforms(pkg.make_camera(), w, h)
        img, bgra = rend.render_host(u, want_rgba=True, want_bgra=True)
        np.testing.assert_array_equal(bgra, oracle.pack_bgra8(img))
        ref = oracle.stages(oracle.activate_records(rec), oracle.camera_uniforms(oracle.default_camera(), w, h))["image"]
        assert np.abs(bgra.astype(int) - oracle.pack_bgra8(ref).astype(int)).max() <= 1
        only8, b8 = rend.render_host(u, want_rgba=False, want_bgra=True)
        assert only8 is None
        np.testing.assert_array_equal(b8, bgra)
        rend.close()
        scene.close()


def test_guard_state_survives_mode_switches_and_frames_in_flight(pkg, oracle, gpu):
    """Mode 3 with three frames in flight and HIP-graph replay renders the same frame as one frame at a time; switching
    3 -> 2 -> 3 changes nothing; with the contractions on, mode 3 runs as mode 1 (documented)."""
    rec = pkg.synth.synth_records(30000, seed=11, kind="A")
    w, h = 400, 300
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    rend.set_exp_mode(3)
    a, _ = rend.render_host(u)
    redo, resolved = rend.stats().blend_redo, rend.stats().blend_resolved
    rend.set_exp_mode(2)
    rend.render_host(u)
    assert rend.stats().blend_redo == 0 and rend.stats().blend_resolved == 0  # the exact mode has nothing to guard
    rend.set_exp_mode(3)
    rend.set_frames_in_flight(3)
    rend.set_graph_mode(True)
    for _ in range(7):
        b, _ = rend.render_host(u)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert rend.stats().blend_redo == redo and rend.stats().blend_resolved == resolved
    rend.set_graph_mode(False)
    rend.set_blend_contraction(True)
    c3, _ = rend.render_host(u)
    rend.set_exp_mode(1)
    c1, _ = rend.render_host(u)
    np.testing.assert_array_equal(c3.view(np.uint32), c1.view(np.uint32))
    rend.close()
    scene.close()


def test_lockstep_with_quadrants_that_end_early_abandon_and_walk_on(pkg, oracle, gpu):
    """The lockstep barrier (gs_blend.hip: one s_barrier per chunk, the tile's four waves together) sits in a loop that each wave of a
    tile may leave at a different chunk: a quadrant whose pixels have all saturated stops, a quadrant the guard abandons re-renders
    itself exactly while its siblings walk on (round-5 advisor finding: correctness rests on ended waves no longer counting for the
    barrier).  One scene with all three in the same tiles: the adversarial stack (guard replays and abandoned quadrants, 64 layers =
    254 layers = four or five chunks with the plugs' entries) and ~60 opaque plugs of ten tight layers each in front of it, scattered so that some
    quadrants are covered whole (dead after a few entries), some in part, some not at all.  Lockstep pinned ON must give the frames
    of lockstep OFF bit for bit in both modes, the exact mode the reference's frame, the default mode its decisions (<= 1e-5)."""
    w = h = 512
    rng = np.random.default_rng(77)
    stack = adversarial_vertices(shift=(0.05, -0.1), strong=0.9006, weak_layers=250, seed=3)
    f = w / (2 * np.tan(np.radians(45.0) / 2))
    plugs = []
    for _ in range(60):
        cx, cy = rng.uniform(-1.9, 1.9, 2)
        sigma_px = rng.uniform(5.0, 14.0)
        for layer in range(10):
            d = 4.0 + 0.01 * layer + rng.uniform(0, 0.005)
            plugs.append((cx * d / 4.0, cy * d / 4.0, -d, sigma_px * d / f))
    plugs = np.array(plugs)
    pv = _vertices(plugs[:, :3], np.repeat(plugs[:, 3:4], 3, axis=1), np.full(len(plugs), 0.999, np.float32), rng.uniform(0.1, 0.9, (len(plugs), 3)))
    verts = np.concatenate([pv, stack])
    u_ref, ref = _reference(oracle, verts, w, h)
    scene = pkg.Scene.from_vertices(verts, device=0)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    frames = {}
    for lockstep in (0, 1):
        rend = pkg.Renderer(scene)
        rend.set_blend_lockstep(lockstep)
        rend.set_exp_mode(2)
        frames[(lockstep, 2)] = rend.render_host(u)[0]
        if lockstep == 0:
            compare_stages(pkg, rend, u, ref)
        worst, redo, resolved = assert_guarded_close(rend, u, ref["image"], label=f"plugs + stack, lockstep {lockstep}: guarded blend")
        frames[(lockstep, 3)] = rend.render_host(u)[0]
        print(f"lockstep {lockstep}: guarded max|d| {worst:.3g}, {resolved} decisions resolved exactly, {redo} quadrants re-rendered")
        assert resolved > 0 and redo > 0, "the scene must make the guard replay AND abandon (else it does not test the mix)"
        rend.close()
    assert_images_identical(frames[(0, 2)], ref["image"], label="exp mode 2, lockstep off")
    for mode in (2, 3):
        np.testing.assert_array_equal(frames[(0, mode)].view(np.uint32), frames[(1, mode)].view(np.uint32))
    # ... and the premise itself: tiles exist whose four waves leave the chunk loop at different chunks
    end_chunk = _quadrant_end_chunks(ref, w, h)
    per_tile = end_chunk.reshape(h // 16, 2, w // 16, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
    mixed = int((per_tile.max(axis=1) > per_tile.min(axis=1)).sum())
    print(f"tiles whose quadrants end in different chunks: {mixed} of {len(per_tile)}; end chunks {np.bincount(end_chunk.ravel())}")
    assert mixed >= 20, "too few tiles mix early-ending and long-walking quadrants"
    scene.close()


def _quadrant_end_chunks(ref, w, h):
    """float64 re-trace (per tile list): for every 8 x 8 quadrant the 64-entry chunk of its tile's list in which its last pixel breaks
    (render.comp:83), or the list's last chunk if one never does -- where the quadrant's wave leaves the blend's chunk loop."""
    attr, bounds, payload = ref["attr"], ref["boundaries"], ref["sorted_payload"]
    tx, ty = (w + 15) // 16, (h + 15) // 16
    out = np.zeros((2 * ty, 2 * tx), np.int64)
    for t in range(tx * ty):
        ids = payload[bounds[2 * t]:bounds[2 * t + 1]]
        n = len(ids)
        if n == 0:
            continue
        x0, y0 = (t % tx) * 16, (t // tx) * 16
        ys, xs = np.mgrid[y0:y0 + 16, x0:x0 + 16].astype(np.float64)
        co = attr["conic_opacity"][ids].astype(np.float64)
        uv = attr["uv"][ids].astype(np.float64)
        dx = uv[:, 0][:, None, None] - xs[None]
        dy = uv[:, 1][:, None, None] - ys[None]
        power = -0.5 * (co[:, 0][:, None, None] * dx * dx + co[:, 2][:, None, None] * dy * dy) - co[:, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, co[:, 3][:, None, None] * np.exp(np.minimum(power, 0)))
        kept = (power <= 0) & (alpha >= 1 / 255)
        T = np.cumprod(np.where(kept, 1 - alpha, 1.0), axis=0)
        broke = kept & (T < 1e-4)
        first = np.where(broke.any(axis=0), broke.argmax(axis=0), n - 1)   # [16, 16] entry at which the pixel stops
        q = first.reshape(2, 8, 2, 8).max(axis=(1, 3))
        out[2 * (t // tx):2 * (t // tx) + 2, 2 * (t % tx):2 * (t % tx) + 2] = q // 64
    return out


def _exactly_centred_vertices(oracle, w, h, n_want=24, seed=5):
    """Vertices whose projected centres land EXACTLY on integer pixels (dx = dy = 0 there, so render.comp:66's `power` is a zero --
    of either sign, depending on the sign of the conic's off-diagonal term).  Found by sweeping adjacent binary32 positions through
    the checker's preprocess until uv is an integer in both coordinates."""
    rng = np.random.default_rng(seed)
    tan = np.tan(np.radians(45.0) / 2)
    found = []
    for _ in range(400):
        if len(found) >= n_want:
            break
        z = float(rng.uniform(2.5, 7.0))
        kx, ky = int(rng.integers(8, w - 8)), int(rng.integers(8, h - 8))
        x0 = np.float32(((2 * kx + 1) / w - 1) * tan * z)
        y0 = np.float32(-((2 * ky + 1) / h - 1) * tan * (h / w) * z)
        xs = (np.arange(-300, 301, dtype=np.int64) + np.int64(x0.view(np.int32))).astype(np.int32).view(np.float32)
        ys = (np.arange(-300, 301, dtype=np.int64) + np.int64(y0.view(np.int32))).astype(np.int32).view(np.float32)
        m = len(xs)
        pos = np.stack([xs, ys, np.full(m, -z, np.float32)], axis=1)
        v = _vertices(pos, np.full((m, 3), 0.05), np.full(m, 0.9, np.float32), np.full((m, 3), 0.5))
        vv = np.ascontiguousarray(v).view(oracle.VERTEX_DT).reshape(-1)
        attr, tiles = oracle.preprocess(vv, oracle.cov3d(vv), oracle.camera_uniforms(oracle.default_camera(), w, h))
        ux = np.nonzero((attr["uv"][:, 0] == kx) & (tiles > 0))[0]
        uy = np.nonzero((attr["uv"][:, 1] == ky) & (tiles > 0))[0]
        if len(ux) and len(uy):
            found.append((xs[ux[0]], ys[uy[0]], -z, kx, ky))
    assert len(found) >= 8, f"only {len(found)} exactly centred positions found"
    f = np.array(found, np.float64)
    n = len(f)
    # anisotropic, rotated about the view axis by angles of both signs: the conic's c01 takes both signs (and ~0)
    ang = rng.uniform(-1.4, 1.4, n)
    ang[:3] = 0.0
    v = _vertices(f[:, :3].astype(np.float32), np.stack([rng.uniform(0.02, 0.08, n), rng.uniform(0.005, 0.02, n), np.full(n, 0.01)], axis=1),
                  rng.uniform(0.3, 0.999, n).astype(np.float32), rng.uniform(0.1, 0.9, (n, 3)))
    v[:, 8] = np.cos(ang / 2)   # rotation quaternion (w, x, y, z): about z
    v[:, 11] = np.sin(ang / 2)
    return v, f[:, 3].astype(int), f[:, 4].astype(int)


def test_pixels_exactly_on_a_splats_centre(pkg, oracle, gpu):
    """render.comp:66 at dx = dy = 0: power is +0 or -0 depending on the sign of the conic's off-diagonal term, and `power > 0` is
    false for both, so the pixel takes alpha = min(0.99, o).  The hand-written pair loops decide :68 and :78 with ONE unsigned compare
    on pn = -power's bit pattern, which rests on pn never being -0 (gs_blend.hip): splats whose centres sit exactly on integer pixels,
    rotated so that c01 takes both signs, must give the reference's frame bit for bit (exp mode 2) and its decisions (default)."""
    w, h = 320, 192
    verts, kx, ky = _exactly_centred_vertices(oracle, w, h)
    u_ref, ref = _reference(oracle, verts, w, h)
    attr = ref["attr"]
    on_centre = (attr["uv"][:, 0] == kx) & (attr["uv"][:, 1] == ky)
    c01 = attr["conic_opacity"][:, 1]
    assert on_centre.sum() >= 8 and (c01[on_centre] > 0).any() and (c01[on_centre] < 0).any()
    scene = pkg.Scene.from_vertices(verts, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    assert u.tobytes() == u_ref.tobytes()
    rend.set_exp_mode(2)
    exact, _ = rend.render_host(u)
    compare_stages(pkg, rend, u, ref)
    assert_images_identical(exact, ref["image"], label="exactly centred splats, exp mode 2")
    assert_guarded_close(rend, u, ref["image"], label="exactly centred splats: guarded blend")
    # the centre pixels did take the splat: each is brighter than black
    assert all(ref["image"][y, x, :3].sum() > 0 for x, y in zip(kx[on_centre], ky[on_centre]))
    rend.close()
    scene.close()
