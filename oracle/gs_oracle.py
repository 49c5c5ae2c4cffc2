"""ctypes binding of oracle/libgs_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VERTEX_DT = np.dtype([("position", "<f4", 4), ("scale_opacity", "<f4", 4), ("rotation", "<f4", 4),
                      ("sh", "<f4", 48)])
ATTR_DT = np.dtype([("conic_opacity", "<f4", 4), ("color_radii", "<f4", 4), ("aabb", "<u4", 4),
                    ("uv", "<f4", 2), ("depth", "<f4"), ("magic", "<u4")])
UNIFORMS_DT = np.dtype([("camera_position", "<f4", 4), ("proj_mat", "<f4", 16), ("view_mat", "<f4", 16),
                        ("width", "<u4"), ("height", "<u4"), ("tan_fovx", "<f4"), ("tan_fovy", "<f4")])
CAMERA_DT = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("fov", "<f4"),
                      ("near_plane", "<f4"), ("far_plane", "<f4")])
assert VERTEX_DT.itemsize == 240 and ATTR_DT.itemsize == 64 and UNIFORMS_DT.itemsize == 160


class Stats(C.Structure):
    _fields_ = [("num_gaussians", C.c_uint64), ("num_visible", C.c_uint64), ("num_instances", C.c_uint64),
                ("ms", C.c_double * 6)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgs_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.gso_exp.restype = C.c_float
        _LIB.gso_exp.argtypes = [C.c_float]
        _LIB.gso_expf_libm.restype = C.c_float
        _LIB.gso_expf_libm.argtypes = [C.c_float]
        _LIB.gso_expf_libm_mismatches.restype = C.c_uint64
        _LIB.gso_expf_device_mismatches.restype = C.c_uint64
        _LIB.gso_expf_monotone_violations.restype = C.c_uint64
        _LIB.gso_alpha_cut.restype = C.c_float
        _LIB.gso_alpha_cut.argtypes = [C.c_float]
        _LIB.gso_render_frame.restype = C.c_int
        _LIB.gso_load_ply.restype = C.c_int
        _LIB.gso_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_camera(position=(0, 0, 0), rotation=(1, 0, 0, 0), fov=45.0, near=0.1, far=1000.0):
    """Renderer::Camera defaults, src/Renderer.h:79-85."""
    cam = np.zeros(1, CAMERA_DT)
    cam["position"] = position
    cam["rotation"] = rotation
    cam["fov"], cam["near_plane"], cam["far_plane"] = fov, near, far
    return cam


def camera_uniforms(cam, width, height):
    out = np.zeros(1, UNIFORMS_DT)
    lib().gso_camera_uniforms(_p(cam), C.c_uint32(width), C.c_uint32(height), _p(out))
    return out


def exp(x):
    x = np.asarray(x, np.float32)
    return np.array([lib().gso_exp(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def activate_records(records):
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 62)
    out = np.zeros(len(records), VERTEX_DT)
    lib().gso_activate_records(_p(records), C.c_uint64(len(records)), _p(out))
    return out


def load_ply(path):
    ptr = C.c_void_p()
    n = C.c_uint64()
    rc = lib().gso_load_ply(path.encode(), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"gso_load_ply({path}) failed: {rc}")
    buf = (C.c_char * (n.value * VERTEX_DT.itemsize)).from_address(ptr.value)
    out = np.frombuffer(buf, VERTEX_DT).copy()
    lib().gso_free(ptr)
    return out


def cov3d(verts):
    out = np.zeros((len(verts), 6), np.float32)
    lib().gso_cov3d(_p(verts), C.c_uint64(len(verts)), _p(out))
    return out


def preprocess(verts, cov, uniforms):
    n = len(verts)
    attr = np.zeros(n, ATTR_DT)
    tiles = np.zeros(n, np.uint32)
    lib().gso_preprocess(_p(verts), _p(cov), C.c_uint64(n), _p(uniforms), _p(attr), _p(tiles))
    return attr, tiles


def inclusive_scan(tiles):
    out = np.zeros_like(tiles)
    lib().gso_inclusive_scan(_p(tiles), C.c_uint64(len(tiles)), _p(out))
    return out


def duplicate(attr, prefix, tile_x):
    d = int(prefix[-1]) if len(prefix) else 0
    keys = np.zeros(d, np.uint64)
    payload = np.zeros(d, np.uint32)
    lib().gso_duplicate(_p(attr), _p(prefix), C.c_uint64(len(attr)), C.c_uint32(tile_x), _p(keys), _p(payload))
    return keys, payload


def sort_pairs(keys, payload):
    keys, payload = keys.copy(), payload.copy()
    lib().gso_sort_pairs(_p(keys), _p(payload), C.c_uint64(len(keys)))
    return keys, payload


def tile_boundary(keys, num_tiles):
    out = np.zeros(2 * num_tiles, np.uint32)
    lib().gso_tile_boundary(_p(keys), C.c_uint64(len(keys)), _p(out), C.c_uint64(num_tiles))
    return out


def render(attr, boundaries, payload, width, height, simd=False):
    """render.comp restated; simd=True: the AVX2 variant used for the CPU baseline (bit-identical by test)."""
    rgba = np.zeros((height, width, 4), np.float32)
    fn = lib().gso_render_simd if simd else lib().gso_render
    fn(_p(attr), _p(boundaries), _p(payload), C.c_uint32(width), C.c_uint32(height), _p(rgba))
    return rgba


def set_contraction(on):
    """render(): False (default) = render.comp:66,87 uncontracted, as the reference's shader text compiled for the CPU
    evaluates them; True = the three FMA contractions of the product's fast blend (gso_set_contraction)."""
    lib().gso_set_contraction(C.c_int(int(bool(on))))


def set_exp_mode(mode):
    """render(): 2 (default) = libm's expf restated (gso_expf_libm), what the reference's shader text compiled for the
    CPU calls; 0 = the pipeline polynomial gso_exp of the product's fast blend."""
    lib().gso_set_exp_mode(C.c_int(int(mode)))


class fast_reading:
    """with oracle.fast_reading(): render() evaluates the product's opt-in FAST blend -- the three FMA contractions GLSL
    permits in render.comp:66,87 and the pipeline-defined polynomial exp (gs_set_blend_contraction(1) + gs_set_exp_mode(0)).
    Outside of it render() is the reference reading: uncontracted, libm's expf -- operation for operation what the
    reference's text compiled for the CPU (oracle/_ref) evaluates, and the product's default."""

    def __enter__(self):
        set_contraction(True)
        set_exp_mode(0)

    def __exit__(self, *a):
        set_contraction(False)
        set_exp_mode(2)


class reading:
    """with oracle.reading(contract, exp_mode): any combination (exp_mode 0 or 2)."""

    def __init__(self, contract, exp_mode):
        self.c, self.e = contract, exp_mode

    def __enter__(self):
        set_contraction(self.c)
        set_exp_mode(self.e)

    def __exit__(self, *a):
        set_contraction(False)
        set_exp_mode(2)


def expf_libm(x):
    x = np.asarray(x, np.float32)
    return np.array([lib().gso_expf_libm(float(v)) for v in x.ravel()], np.float32).reshape(x.shape)


def expf_libm_mismatches(first_bits, count):
    """(# of binary32 bit patterns in [first_bits, first_bits + count) where gso_expf_libm != this machine's expf, first)"""
    bad = C.c_uint32(0)
    n = lib().gso_expf_libm_mismatches(C.c_uint32(first_bits), C.c_uint64(count), C.byref(bad))
    return int(n), int(bad.value)


def expf_device_mismatches(first_bits, count):
    """The same count for gso_expf_device, the operation sequence the HIP kernels use (one binary64 operation fewer)."""
    bad = C.c_uint32(0)
    n = lib().gso_expf_device_mismatches(C.c_uint32(first_bits), C.c_uint64(count), C.byref(bad))
    return int(n), int(bad.value)


def expf_monotone_violations(first_bits, count):
    """# of adjacent binary32 pairs (bits b, b + 1) in the range on which this machine's expf grows as x falls: the alpha cut's premise."""
    return int(lib().gso_expf_monotone_violations(C.c_uint32(first_bits), C.c_uint64(count)))


def alpha_cut(opacity):
    """render.comp:78 as a bound on `power` (gso_alpha_cut): per opacity, the most negative power at which an entry is kept."""
    o = np.ascontiguousarray(np.atleast_1d(opacity), np.float32)
    out = np.empty(o.shape, np.float32)
    lib().gso_alpha_cut_array(o.ctypes.data_as(C.c_void_p), C.c_uint64(o.size), out.ctypes.data_as(C.c_void_p))
    return out


def libm_expf_block_sums(first_bits, count):
    """Checksums of this machine's expf over blocks of 2^20 bit patterns, as the device hook gs_debug_expf_scan forms them."""
    blocks = (count + (1 << 20) - 1) >> 20
    out = np.zeros(blocks, np.uint64)
    lib().gso_libm_expf_block_sums(C.c_uint32(first_bits), C.c_uint64(count), out.ctypes.data_as(C.c_void_p))
    return out


def set_simd_blend(on):
    """render_frame uses the AVX2 blend (CPU baseline timing only; the parity checks use the scalar one)."""
    lib().gso_set_simd_blend(C.c_int(int(on)))


def render_frame(verts, cov, uniforms, want_image=True):
    """Whole frame in Renderer::draw order; returns (rgba or None, Stats)."""
    h, w = int(uniforms["height"][0]), int(uniforms["width"][0])
    rgba = np.zeros((h, w, 4), np.float32) if want_image else None
    st = Stats()
    rc = lib().gso_render_frame(_p(verts), _p(cov), C.c_uint64(len(verts)), _p(uniforms),
                                _p(rgba) if want_image else None, C.byref(st))
    if rc != 0:
        raise MemoryError("gso_render_frame")
    return rgba, st


def pack_bgra8(rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros(rgba.shape[:-1] + (4,), np.uint8)
    lib().gso_pack_bgra8(_p(rgba), C.c_uint64(rgba.size // 4), _p(out))
    return out


def stages(verts, uniforms):
    """All per-stage outputs for parity taps."""
    w, h = int(uniforms["width"][0]), int(uniforms["height"][0])
    tx, ty = (w + 15) // 16, (h + 15) // 16
    cov = cov3d(verts)
    attr, tiles = preprocess(verts, cov, uniforms)
    prefix = inclusive_scan(tiles)
    keys, payload = duplicate(attr, prefix, tx)
    skeys, spayload = sort_pairs(keys, payload)
    bounds = tile_boundary(skeys, tx * ty)
    img = render(attr, bounds, spayload, w, h)
    return dict(cov3d=cov, attr=attr, tiles=tiles, prefix=prefix, keys=keys, payload=payload,
                sorted_keys=skeys, sorted_payload=spayload, boundaries=bounds, image=img)


def num_threads():
    return lib().gso_num_threads()


def set_num_threads(n):
    lib().gso_set_num_threads(C.c_int(n))
