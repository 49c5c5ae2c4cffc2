"""The host-side arithmetic -- GSScene::load's record conversion, Renderer::updateUniforms, Camera::translate -- pinned to the
REFERENCE'S OWN TEXT and to exact arithmetic (no GPU needed: these are host functions of the C ABI).

  * oracle/_ref compiles GSScene.cpp:17-24,37-58, GSScene.h:41-46, Renderer.h:21-29,40-50 and Renderer.cpp:719-754 verbatim against
    oracle/glsl_cpu/glm_stub.hpp (oracle/build_ref.py: host_text_to_cpp).  The product (gs_activate_records, gs_camera_uniforms,
    csrc/host/gs_linalg.h) and the restated oracle must equal that bit for bit.
  * What the stub holds -- the inside of glm::mat4_cast / translate / operator* / inverse / perspective -- is checked a second way,
    by a Python evaluation in EXACT rational arithmetic (fractions.Fraction) with an explicit round-to-nearest-even to binary32
    after every single operation, in the order glm's sources write them.  The three transcendental calls of the path (tan in
    double, atanf, tanf) are taken from the same libm the product links.
"""
import ctypes
import math
import os
import struct
import subprocess
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref(pkg):
    import __graft_entry__ as entry
    r = entry.load_ref()
    if not r.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not mounted")
    return r


def random_cameras(pkg, count, seed):
    rng = np.random.default_rng(seed)
    cams = []
    for k in range(count):
        q = rng.normal(size=4)
        if k % 3:  # most cameras carry a unit quaternion, like the viewer's; every third an arbitrary one
            q /= np.linalg.norm(q)
        pos = rng.normal(size=3) * (10.0 ** rng.integers(-2, 3))
        near = float(10.0 ** rng.uniform(-3, 0))
        cam = pkg.make_camera(position=tuple(pos), rotation=tuple(q), fov=float(rng.uniform(5, 150)), near=near,
                              far=near * float(10.0 ** rng.uniform(1, 5)))
        w, h = int(rng.integers(1, 4097)), int(rng.integers(1, 2305))
        cams.append((cam, w, h))
    cams.append((pkg.make_camera(), 1920, 1080))
    cams.append((pkg.make_camera(position=(0.0, -0.0, 3.5), rotation=(0.0, 1.0, 0.0, 0.0)), 3840, 2160))
    return cams


def test_update_uniforms_equals_the_reference_text(pkg, oracle, ref):
    for cam, w, h in random_cameras(pkg, 1000, seed=7):
        want = ref.update_uniforms(cam, w, h).tobytes()
        assert pkg.camera_uniforms(cam, w, h).tobytes() == want, (cam, w, h)       # the product: gs_camera_uniforms
        assert oracle.camera_uniforms(cam, w, h).tobytes() == want, (cam, w, h)    # the restated oracle


def test_record_conversion_equals_the_reference_text(pkg, oracle, ref):
    rng = np.random.default_rng(11)
    rec = pkg.synth.synth_records(50_000, seed=3, kind="T")
    extra = rng.normal(size=(20_000, 62)).astype(np.float32)
    extra[:, 55:58] *= 4.0                                     # scales exp() to 1e-7 .. 1e7
    extra[:4000, 54] = rng.uniform(-90, 90, 4000)              # opacities whose sigmoid saturates / underflows
    extra[4000:4100, 58:62] *= np.float32(1e-20)               # quaternions whose squared norm is denormal
    extra[4100:4200, 58:62] *= np.float32(1e18)                # ... or overflows
    rec = np.concatenate([rec, extra])
    want = ref.load_records(rec)
    got, port = pkg.activate_records(rec), oracle.activate_records(rec)
    # bit patterns, NaN included (0 * inf from a vanishing quaternion must come out the same way)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(np.ascontiguousarray(port).view(np.uint32).reshape(want.shape), want.view(np.uint32))


def test_camera_translate_equals_the_reference_text(pkg, ref, tmp_path):
    """Renderer::Camera::translate (Renderer.h:47-49) as lib3dgs_cpp's logMovement evaluates it (csrc/host/gs_linalg.h)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "gs_linalg.h"\nextern "C" void tr(const float* c, const float* t, float* o) {\n'
                   '  gs::quat q{c[3], c[4], c[5], c[6]}; gs::vec3 p{c[0], c[1], c[2]};\n'
                   '  gs::vec3 r = p + q * gs::vec3{t[0], t[1], t[2]}; o[0] = r.x; o[1] = r.y; o[2] = r.z; }\n')
    so = tmp_path / "t.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "3dgs.cpp_amd", "csrc", "host"),
                           str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    rng = np.random.default_rng(5)
    for cam, _, _ in random_cameras(pkg, 300, seed=9):
        t = rng.normal(size=3).astype(np.float32)
        got = np.zeros(3, np.float32)
        lib.tr(ref._cam10(cam).ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p))
        assert got.tobytes() == ref.camera_translate(cam, t).tobytes()


# ---------------------------------------------------------------- exact arithmetic
def f32(x):
    """Round a Fraction to the nearest binary32 (ties to even), as a Fraction.  The values of this path stay far from overflow."""
    if x == 0:
        return Fraction(0)
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)                      # denormals share the exponent of the smallest normal
    q = a / Fraction(2) ** (e - 23)       # the significand in units of the last place
    n = q.numerator // q.denominator
    r = q - n
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return s * n * Fraction(2) ** (e - 23)


def F(v):
    return Fraction(float(np.float32(v)))


def bits(fr):
    return struct.unpack("<I", struct.pack("<f", float(fr)))[0]


class X:  # binary32 arithmetic with one exact rounding per operation
    add = staticmethod(lambda a, b: f32(a + b))
    sub = staticmethod(lambda a, b: f32(a - b))
    mul = staticmethod(lambda a, b: f32(a * b))
    div = staticmethod(lambda a, b: f32(a / b))


def exact_uniforms(cam, w, h, libm):
    px, py, pz = (F(v) for v in cam["position"][0])
    qw, qx, qy, qz = (F(v) for v in cam["rotation"][0])
    one, two, zero = Fraction(1), Fraction(2), Fraction(0)
    # gtc/quaternion.inl mat3_cast
    qxx, qyy, qzz = X.mul(qx, qx), X.mul(qy, qy), X.mul(qz, qz)
    qxz, qxy, qyz = X.mul(qx, qz), X.mul(qx, qy), X.mul(qy, qz)
    qwx, qwy, qwz = X.mul(qw, qx), X.mul(qw, qy), X.mul(qw, qz)
    rot = [[X.sub(one, X.mul(two, X.add(qyy, qzz))), X.mul(two, X.add(qxy, qwz)), X.mul(two, X.sub(qxz, qwy)), zero],
           [X.mul(two, X.sub(qxy, qwz)), X.sub(one, X.mul(two, X.add(qxx, qzz))), X.mul(two, X.add(qyz, qwx)), zero],
           [X.mul(two, X.add(qxz, qwy)), X.mul(two, X.sub(qyz, qwx)), X.sub(one, X.mul(two, X.add(qxx, qyy))), zero],
           [zero, zero, zero, one]]
    ident = [[one if r == c else zero for r in range(4)] for c in range(4)]
    # ext/matrix_transform.inl translate: Result[3] = m[0] v0 + m[1] v1 + m[2] v2 + m[3]
    tr = [col[:] for col in ident]
    tr[3] = [X.add(X.add(X.add(X.mul(ident[0][r], px), X.mul(ident[1][r], py)), X.mul(ident[2][r], pz)), ident[3][r]) for r in range(4)]

    def matmul(a, b):  # type_mat4x4.inl: Result[j] = A0 Bj[0] + A1 Bj[1] + A2 Bj[2] + A3 Bj[3]
        return [[X.add(X.add(X.add(X.mul(a[0][r], b[j][0]), X.mul(a[1][r], b[j][1])), X.mul(a[2][r], b[j][2])), X.mul(a[3][r], b[j][3]))
                 for r in range(4)] for j in range(4)]
    m = matmul(tr, rot)

    def coef(a, b, c, d):
        return X.sub(X.mul(a, b), X.mul(c, d))
    C00, C02, C03 = coef(m[2][2], m[3][3], m[3][2], m[2][3]), coef(m[1][2], m[3][3], m[3][2], m[1][3]), coef(m[1][2], m[2][3], m[2][2], m[1][3])
    C04, C06, C07 = coef(m[2][1], m[3][3], m[3][1], m[2][3]), coef(m[1][1], m[3][3], m[3][1], m[1][3]), coef(m[1][1], m[2][3], m[2][1], m[1][3])
    C08, C10, C11 = coef(m[2][1], m[3][2], m[3][1], m[2][2]), coef(m[1][1], m[3][2], m[3][1], m[1][2]), coef(m[1][1], m[2][2], m[2][1], m[1][2])
    C12, C14, C15 = coef(m[2][0], m[3][3], m[3][0], m[2][3]), coef(m[1][0], m[3][3], m[3][0], m[1][3]), coef(m[1][0], m[2][3], m[2][0], m[1][3])
    C16, C18, C19 = coef(m[2][0], m[3][2], m[3][0], m[2][2]), coef(m[1][0], m[3][2], m[3][0], m[1][2]), coef(m[1][0], m[2][2], m[2][0], m[1][2])
    C20, C22, C23 = coef(m[2][0], m[3][1], m[3][0], m[2][1]), coef(m[1][0], m[3][1], m[3][0], m[1][1]), coef(m[1][0], m[2][1], m[2][0], m[1][1])
    Fac = [[C00, C00, C02, C03], [C04, C04, C06, C07], [C08, C08, C10, C11], [C12, C12, C14, C15], [C16, C16, C18, C19], [C20, C20, C22, C23]]
    Vec = [[m[1][k], m[0][k], m[0][k], m[0][k]] for k in range(4)]

    def inv(va, fa, vb, fb, vc, fc):  # Va * Fa - Vb * Fb + Vc * Fc, component-wise, left to right
        return [X.add(X.sub(X.mul(va[k], fa[k]), X.mul(vb[k], fb[k])), X.mul(vc[k], fc[k])) for k in range(4)]
    Inv = [inv(Vec[1], Fac[0], Vec[2], Fac[1], Vec[3], Fac[2]), inv(Vec[0], Fac[0], Vec[2], Fac[3], Vec[3], Fac[4]),
           inv(Vec[0], Fac[1], Vec[1], Fac[3], Vec[3], Fac[5]), inv(Vec[0], Fac[2], Vec[1], Fac[4], Vec[2], Fac[5])]
    sign_a, sign_b = [1, -1, 1, -1], [-1, 1, -1, 1]
    Inverse = [[X.mul(Inv[c][k], Fraction((sign_a if c % 2 == 0 else sign_b)[k])) for k in range(4)] for c in range(4)]
    dot0 = [X.mul(m[0][k], Inverse[k][0]) for k in range(4)]
    dot1 = X.add(X.add(dot0[0], dot0[1]), X.add(dot0[2], dot0[3]))
    ood = X.div(one, dot1)
    view = [[X.mul(Inverse[c][r], ood) for r in range(4)] for c in range(4)]
    # Renderer.cpp:730-735: tan in double on radians(fov) / 2.0, narrowed; the float chain after it
    fov = F(cam["fov"][0])
    rad = X.mul(fov, F(np.float32(0.01745329251994329576923690768489)))
    tan_fovx = F(np.float32(libm.tan(float(rad) / 2.0)))
    tan_fovy = X.div(X.mul(tan_fovx, Fraction(h)), Fraction(w)) if float(np.float32(h)) == h and float(np.float32(w)) == w else None
    fovy = X.mul(F(libm.atanf(ctypes.c_float(float(tan_fovy)))), two)
    aspect = X.div(Fraction(w), Fraction(h))
    tan_half = F(libm.tanf(ctypes.c_float(float(X.div(fovy, two)))))
    near, far = F(cam["near_plane"][0]), F(cam["far_plane"][0])
    persp = [[zero] * 4 for _ in range(4)]
    persp[0][0] = X.div(one, X.mul(aspect, tan_half))
    persp[1][1] = X.div(one, tan_half)
    persp[2][2] = X.div(-X.add(far, near), X.sub(far, near))
    persp[2][3] = -one
    persp[3][2] = X.div(-X.mul(X.mul(two, far), near), X.sub(far, near))
    proj = matmul(persp, view)
    for c in range(4):
        view[c][1], view[c][2], proj[c][1] = -view[c][1], -view[c][2], -proj[c][1]
    words = [bits(px), bits(py), bits(pz), bits(one)]
    words += [bits(proj[c][r]) for c in range(4) for r in range(4)] + [bits(view[c][r]) for c in range(4) for r in range(4)]
    return words + [w, h, bits(tan_fovx), bits(tan_fovy)]


def test_update_uniforms_in_exact_rational_arithmetic(pkg):
    libm = ctypes.CDLL("libm.so.6")
    libm.tan.restype, libm.tan.argtypes = ctypes.c_double, [ctypes.c_double]
    libm.atanf.restype, libm.atanf.argtypes = ctypes.c_float, [ctypes.c_float]
    libm.tanf.restype, libm.tanf.argtypes = ctypes.c_float, [ctypes.c_float]
    checked = 0
    for cam, w, h in random_cameras(pkg, 150, seed=21):
        got = np.frombuffer(pkg.camera_uniforms(cam, w, h).tobytes(), np.uint32)
        want = np.array(exact_uniforms(cam, w, h, libm), np.uint64)
        # -0.0 and +0.0 compare equal here: a product with an exact zero carries the sign of its operands, which the rational form drops
        same = (got == want) | (((got | want) & 0x7FFFFFFF) == 0)
        assert same.all(), (cam, w, h, np.nonzero(~same)[0])
        checked += 1
    assert checked > 100
