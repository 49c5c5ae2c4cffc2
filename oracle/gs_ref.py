"""ctypes binding of oracle/_ref/libgs_ref.so -- the reference's own shader text compiled for the CPU
(oracle/build_ref.py), the radix sort (sort/hist.comp, sort/sort.comp) included.  TEST INFRASTRUCTURE ONLY, like gs_oracle.py: tests pin the restated oracle against it.

The library is built where /root/reference is mounted (the CPU container) and travels to the GPU box as a
prebuilt file; available() says whether it is there.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

from gs_oracle import ATTR_DT, VERTEX_DT  # same 240-byte / 64-byte records (common.glsl:35-49)

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libgs_ref.so")
_LIB = None


def build(reference="/root/reference"):
    """(Re)build from the reference tree if it is mounted; otherwise keep the prebuilt library."""
    if os.path.isdir(os.path.join(reference, "src", "shaders")):
        subprocess.check_call([sys.executable, os.path.join(_HERE, "build_ref.py"), "--reference", reference],
                              stdout=subprocess.DEVNULL)
    return os.path.exists(_PATH)


def available():
    return os.path.exists(_PATH) or build()


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libgs_ref.so is missing and /root/reference is not mounted")
        _LIB = C.CDLL(_PATH)
        _LIB.gsr_prefix_sum.restype = C.c_int
        _LIB.gsr_sources.restype = C.c_char_p
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sources():
    """The reference files (with sha256) whose text is compiled into the library."""
    return lib().gsr_sources().decode()


def cov3d(verts, scale_factor=1.0):
    out = np.zeros((len(verts), 6), np.float32)
    lib().gsr_precomp_cov3d(_p(verts), C.c_uint64(len(verts)), C.c_float(scale_factor), _p(out))
    return out


def preprocess(verts, cov, uniforms):
    n = len(verts)
    attr = np.zeros(n, ATTR_DT)  # the reference's buffer is not cleared either; zeros make culled records comparable
    tiles = np.zeros(n, np.uint32)
    lib().gsr_preprocess(_p(verts), _p(cov), C.c_uint64(n), _p(uniforms), _p(attr), _p(tiles))
    return attr, tiles


def inclusive_scan(tiles):
    """prefix_sum.comp run for all ceil(log2 N) + 1 timesteps (Renderer.cpp:497-514)."""
    ping = tiles.copy()  # vkCmdCopyBuffer tiles_overlap -> ping, Renderer.cpp:489-490
    pong = np.zeros_like(tiles)
    if len(tiles) == 0:
        return ping
    which = lib().gsr_prefix_sum(_p(ping), _p(pong), C.c_uint64(len(tiles)))
    return pong if which else ping


def duplicate(attr, prefix, tile_x):
    d = int(prefix[-1]) if len(prefix) else 0
    keys = np.zeros(d, np.uint64)
    payload = np.zeros(d, np.uint32)
    lib().gsr_preprocess_sort(_p(attr), _p(prefix), C.c_uint64(len(attr)), C.c_uint32(tile_x), _p(keys), _p(payload),
                              C.c_uint64(d))
    return keys, payload


def sort_pairs(keys, payload):
    keys, payload = keys.copy(), payload.copy()
    lib().gsr_sort_pairs(_p(keys), _p(payload), C.c_uint64(len(keys)))
    return keys, payload


def tile_boundary(keys, num_tiles):
    out = np.zeros(2 * num_tiles, np.uint32)
    lib().gsr_tile_boundary(_p(keys), C.c_uint64(len(keys)), _p(out), C.c_uint64(num_tiles))
    return out


def render(attr, boundaries, payload, width, height):
    rgba = np.zeros((height, width, 4), np.float32)
    lib().gsr_render(_p(attr), C.c_uint64(len(attr)), _p(boundaries), C.c_uint64(len(boundaries) // 2), _p(payload),
                     C.c_uint64(len(payload)), C.c_uint32(width), C.c_uint32(height), _p(rgba))
    return rgba


# Up to this many instances stages() sorts with the reference's own radix-sort text (sort/hist.comp + sort/sort.comp on the CPU
# workgroup emulation: ~1.7 s per million keys); beyond it with std::stable_sort, which the text is pinned to
# (tests/test_oracle_vs_ref.py::test_the_reference_radix_sort_text_*).  Config B (4.9 M instances) is inside.
TEXT_SORT_MAX = int(os.environ.get("GS_REF_TEXT_SORT_MAX", 6_000_000))


def stages(verts, uniforms, cov=None, text_sort=None):
    """Same dictionary as gs_oracle.stages, every stage computed by the reference's shader text.
    cov: the load-time precomp_cov3d result, when the caller times the per-frame passes only.
    text_sort: True = the sort is the reference's radix-sort text too, False = std::stable_sort, None = by TEXT_SORT_MAX."""
    w, h = int(uniforms["width"][0]), int(uniforms["height"][0])
    tx, ty = (w + 15) // 16, (h + 15) // 16
    if cov is None:
        cov = cov3d(verts)
    attr, tiles = preprocess(verts, cov, uniforms)
    prefix = inclusive_scan(tiles)
    keys, payload = duplicate(attr, prefix, tx)
    if text_sort is None:
        text_sort = len(keys) <= TEXT_SORT_MAX
    skeys, spayload = radix_sort_pairs(keys, payload) if text_sort else sort_pairs(keys, payload)
    bounds = tile_boundary(skeys, tx * ty)
    img = render(attr, bounds, spayload, w, h)
    return dict(cov3d=cov, attr=attr, tiles=tiles, prefix=prefix, keys=keys, payload=payload,
                sorted_keys=skeys, sorted_payload=spayload, boundaries=bounds, image=img, text_sort=bool(text_sort))


def radix_sort_pairs(keys, payload, blocks_per_workgroup=32, subgroup_size=32):
    """The reference's OWN radix sort: sort/hist.comp + sort/sort.comp run eight times with Renderer.cpp:598-629's grid, push
    constants and buffer ping-pong (blocks_per_workgroup: Renderer.h:134-138; subgroup_size: sort.comp:46 assumes 32)."""
    keys, payload = np.ascontiguousarray(keys, np.uint64).copy(), np.ascontiguousarray(payload, np.uint32).copy()
    lib().gsr_radix_sort_pairs.restype = C.c_int
    rc = lib().gsr_radix_sort_pairs(_p(keys), _p(payload), C.c_uint64(len(keys)), C.c_uint32(blocks_per_workgroup), C.c_uint32(subgroup_size))
    if rc != 0:
        raise ValueError("gsr_radix_sort_pairs: element count beyond the shaders' 32-bit range")
    return keys, payload


# ---- the reference's HOST arithmetic, its own text compiled against glsl_cpu/glm_stub.hpp (build_ref.py: host_text_to_cpp) ----
def load_records(records):
    """GSScene::load's conversion loop (GSScene.cpp:37-58), verbatim: (n, 62) PLY records -> (n, 60) GSScene::Vertex floats."""
    records = np.ascontiguousarray(records, np.float32).reshape(-1, 62)
    out = np.zeros((len(records), 60), np.float32)
    lib().gsr_load_records(_p(records), C.c_uint64(len(records)), _p(out))
    return out


def _cam10(cam):
    """gs_camera (position[3], rotation wxyz, fov, near, far) as ten floats."""
    c = np.zeros(10, np.float32)
    c[0:3] = np.asarray(cam["position"], np.float32).reshape(-1)[:3]
    c[3:7] = np.asarray(cam["rotation"], np.float32).reshape(-1)[:4]
    c[7], c[8], c[9] = float(cam["fov"][0]), float(cam["near_plane"][0]), float(cam["far_plane"][0])
    return c


def update_uniforms(cam, width, height):
    """Renderer::updateUniforms (Renderer.cpp:719-754), verbatim: the 160-byte uniform block as raw bytes."""
    out = np.zeros(160, np.uint8)
    lib().gsr_update_uniforms(_p(_cam10(cam)), C.c_uint32(width), C.c_uint32(height), _p(out))
    return out


def camera_translate(cam, t):
    """Renderer::Camera::translate (Renderer.h:47-49), verbatim: the new position."""
    out = np.zeros(3, np.float32)
    tt = np.asarray(t, np.float32)
    lib().gsr_camera_translate(_p(_cam10(cam)), _p(tt), _p(out))
    return out
