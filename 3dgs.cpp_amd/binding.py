"""ctypes binding of libgs3d_hip.so (include/gs3d_hip.h) for tests and bench.py.

Host-language note: the reference is a C++ library, so the host-side mirror of its API is C++
(csrc/host/, include/3dgs/).  This module is only the Python harness over the same C ABI: it adds
no arithmetic and never falls back to a CPU path -- if the HIP library or a GPU is missing, calls
raise GsError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GS3D_HIP_LIB", os.path.join(_HERE, "libgs3d_hip.so"))  # override: instrumented builds

UNIFORMS_DT = np.dtype([("camera_position", "<f4", 4), ("proj_mat", "<f4", 16), ("view_mat", "<f4", 16),
                        ("width", "<u4"), ("height", "<u4"), ("tan_fovx", "<f4"), ("tan_fovy", "<f4")])
CAMERA_DT = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 4), ("fov", "<f4"),
                      ("near_plane", "<f4"), ("far_plane", "<f4")])
VERTEX_FLOATS = 60
RECORD_FLOATS = 62
BLOB_PLANES = 59


# the device code of libgs3d_hip.so: one translation unit per stage + the shared device headers
KERNEL_SOURCES = ("gs_device.h", "gs_bin.h", "gs_scene.hip", "gs_preprocess.hip", "gs_radix.hip", "gs_bin_l1.hip", "gs_bin_l2.hip",
                  "gs_blend.hip", "gs_kernels.h")


def library_source_hash():
    """sha256 over the KERNEL sources (comments and whitespace stripped) and the build flags libgs3d_hip.so is built from:
    ties a committed rocprofv3 counter file to the kernels it was collected on (bench.py refuses counters of other
    kernels; host-side or comment-only changes do not move a kernel's instruction or byte counts)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES + ("Makefile",):
        with open(os.path.join(_HERE, "csrc", name), "r") as f:
            text = f.read()
        if name != "Makefile":
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
        else:
            text = re.sub(r"#[^\n]*", " ", text)
        h.update(name.encode() + b"\0" + " ".join(text.split()).encode())
    return h.hexdigest()


STAGES = dict(tiles=(0, np.uint32), depth=(1, np.float32), radius=(2, np.float32), aabb=(3, np.uint16),
              conic_opacity=(4, np.float32), uv_rg=(5, np.float32), b=(6, np.float32),
              depth_order=(7, np.uint32), alpha_cut=(8, np.float32), sorted_tile=(11, np.uint32), sorted_gid=(12, np.uint32),
              ranges=(13, np.uint32), lists_raw=(14, np.uint32), ranges_raw=(15, np.uint32))

# every symbol include/gs3d_hip.h declares
SYMBOLS = ["gs_last_error", "gs_device_count", "gs_read_ply", "gs_activate_records", "gs_scene_load_ply", "gs_scene_from_records",
           "gs_scene_from_vertices", "gs_scene_from_device_blob", "gs_scene_blob_floats", "gs_scene_blob",
           "gs_scene_num_vertices", "gs_scene_quantize_sh", "gs_scene_sh_bits", "gs_scene_download_vertex_range",
           "gs_scene_download_vertices", "gs_scene_download_cov3d",
           "gs_scene_destroy", "gs_renderer_create", "gs_renderer_destroy", "gs_camera_uniforms",
           "gs_render", "gs_render_host", "gs_synchronize", "gs_set_timing", "gs_set_frames_in_flight", "gs_set_sort_path", "gs_set_exp_mode", "gs_set_graph_mode", "gs_set_blend_lockstep", "gs_get_blend_lockstep", "gs_set_blend_contraction", "gs_get_timing_totals",
           "gs_get_frame_intervals", "gs_get_stats", "gs_poll_stats", "gs_debug_download", "gs_debug_expf_scan", "gs_renderer_stream",
           "gs_dist_unique_id", "gs_dist_create", "gs_dist_rank", "gs_dist_world", "gs_dist_pose_count",
           "gs_dist_broadcast_scene", "gs_dist_broadcast_scene_ex", "gs_dist_verify", "gs_dist_destroy"]


class FrameStats(C.Structure):
    _fields_ = [("num_gaussians", C.c_uint64), ("num_visible", C.c_uint64), ("num_instances", C.c_uint64),
                ("instance_capacity", C.c_uint64), ("ms_preprocess", C.c_float), ("ms_prefix_sum", C.c_float),
                ("ms_preprocess_sort", C.c_float), ("ms_sort", C.c_float), ("ms_tile_boundary", C.c_float),
                ("ms_render", C.c_float), ("ms_total", C.c_float), ("retries", C.c_uint32),
                ("num_bin_entries", C.c_uint32), ("max_bin_entries", C.c_uint32), ("sort_path", C.c_uint32),
                ("bin_tiles", C.c_uint32), ("sort_level", C.c_uint32), ("blend_redo", C.c_uint32), ("blend_resolved", C.c_uint32), ("pad_", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def debug_expf_scan(first_bits, count, device=0):
    """Test hook gs_debug_expf_scan: (block checksums of the kernels' expf over the bit patterns, the guard's measured premise)."""
    blocks = (count + (1 << 20) - 1) >> 20
    sums = np.zeros(blocks, np.uint64)
    guard = np.zeros(4, np.float64)
    _check(lib().gs_debug_expf_scan(C.c_int(device), C.c_uint32(first_bits), C.c_uint64(count), _p(sums), C.c_uint64(blocks), _p(guard)))
    return sums, guard


class GsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


_LIB = None


def lib():
    """Load libgs3d_hip.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GsError(-3, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950)")
        L = C.CDLL(LIB_PATH)
        L.gs_last_error.restype = C.c_char_p
        L.gs_scene_num_vertices.restype = C.c_uint64
        L.gs_scene_num_vertices.argtypes = [C.c_void_p]
        L.gs_scene_blob_floats.restype = C.c_uint64
        L.gs_scene_blob_floats.argtypes = [C.c_uint64]
        L.gs_renderer_stream.restype = C.c_void_p
        L.gs_renderer_stream.argtypes = [C.c_void_p]
        L.gs_scene_destroy.argtypes = [C.c_void_p]
        L.gs_renderer_destroy.argtypes = [C.c_void_p]
        L.gs_scene_destroy.restype = None
        L.gs_renderer_destroy.restype = None
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise GsError(rc, lib().gs_last_error().decode(errors="replace"))


def _p(a):
    """Pointer to a numpy array's data for the duration of a call (the caller holds the array).  NOT `a.ctypes.data_as(...)`: that
    builds a reference cycle per call (the pointer object keeps the array, the array's ctypes helper keeps the pointer), i.e. two
    objects of cyclic garbage per gs_render -- enough to trigger Python's full (generation-2) collection once per ~2 000 frames,
    which in a process that has imported torch walks ~170 000 objects and holds the frame loop for 35-45 ms: the "once per run"
    stall of bench.py's rounds 2-4 (profiles/r05_stall_hunt.txt).

    CONTRACT: the returned pointer does NOT keep `a` alive.  Pass a NAMED array that outlives the C call -- never a temporary
    (`_p(np.ascontiguousarray(x))` would hand C a dangling pointer).  What can be checked here is: an ndarray, C-contiguous."""
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("_p() takes a C-contiguous numpy array that the caller keeps alive for the duration of the call")
    return C.c_void_p(a.ctypes.data)


def device_count():
    n = C.c_int(0)
    _check(lib().gs_device_count(C.byref(n)))
    return n.value


def make_camera(position=(0, 0, 0), rotation=(1, 0, 0, 0), fov=45.0, near=0.1, far=1000.0):
    """Renderer::Camera with the reference defaults (Renderer.h:79-85)."""
    cam = np.zeros(1, CAMERA_DT)
    cam["position"] = position
    cam["rotation"] = rotation
    cam["fov"], cam["near_plane"], cam["far_plane"] = fov, near, far
    return cam


def camera_uniforms(cam, width, height):
    out = np.zeros(1, UNIFORMS_DT)
    _check(lib().gs_camera_uniforms(_p(cam), C.c_uint32(width), C.c_uint32(height), _p(out)))
    return out


def activate_records(records):
    """GSScene::load's record conversion on the host (no GPU needed): (n, 62) -> (n, 60)."""
    records = np.ascontiguousarray(records, np.float32).reshape(-1, RECORD_FLOATS)
    out = np.zeros((len(records), VERTEX_FLOATS), np.float32)
    _check(lib().gs_activate_records(_p(records), C.c_uint64(len(records)), _p(out)))
    return out


def read_ply(path):
    """loadPlyHeader + payload (host only): raw (n, 62) records."""
    n = C.c_uint64()
    _check(lib().gs_read_ply(os.fsencode(path), None, C.c_uint64(0), C.byref(n)))
    out = np.zeros((n.value, RECORD_FLOATS), np.float32)
    _check(lib().gs_read_ply(os.fsencode(path), _p(out), C.c_uint64(n.value), C.byref(n)))
    return out


class Scene:
    """GSScene counterpart (SoA in HBM)."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keepalive = keepalive

    @classmethod
    def load_ply(cls, path, device=0):
        h = C.c_void_p()
        _check(lib().gs_scene_load_ply(os.fsencode(path), C.c_int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_records(cls, records, device=0):
        records = np.ascontiguousarray(records, np.float32).reshape(-1, RECORD_FLOATS)
        h = C.c_void_p()
        _check(lib().gs_scene_from_records(_p(records), C.c_uint64(len(records)), C.c_int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_vertices(cls, vertices, device=0):
        """vertices: (n, 60) float32 or the structured GSScene::Vertex array."""
        v = np.ascontiguousarray(vertices).view(np.float32).reshape(-1, VERTEX_FLOATS)
        h = C.c_void_p()
        _check(lib().gs_scene_from_vertices(_p(v), C.c_uint64(len(v)), C.c_int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_device_blob(cls, ptr, n, device=0, keepalive=None):
        h = C.c_void_p()
        _check(lib().gs_scene_from_device_blob(C.c_void_p(ptr), C.c_uint64(n), C.c_int(device), C.byref(h)))
        return cls(h, keepalive)

    @property
    def num_vertices(self):
        return lib().gs_scene_num_vertices(self._h)

    def blob(self):
        """(device pointer, float count) of the packed SoA blob."""
        p = C.c_void_p()
        n = C.c_uint64()
        _check(lib().gs_scene_blob(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def download_vertices(self):
        out = np.zeros((self.num_vertices, VERTEX_FLOATS), np.float32)
        _check(lib().gs_scene_download_vertices(self._h, _p(out)))
        return out

    def quantize_sh(self):
        """Round the SH coefficients to binary16 storage (opt-in; gs_scene_quantize_sh)."""
        _check(lib().gs_scene_quantize_sh(self._h))

    @property
    def sh_bits(self):
        return lib().gs_scene_sh_bits(self._h)

    def download_vertex_range(self, first, count):
        out = np.zeros((count, VERTEX_FLOATS), np.float32)
        _check(lib().gs_scene_download_vertex_range(self._h, C.c_uint64(first), C.c_uint64(count), _p(out)))
        return out

    def download_cov3d(self):
        out = np.zeros((self.num_vertices, 6), np.float32)
        _check(lib().gs_scene_download_cov3d(self._h, _p(out)))
        return out

    def close(self):
        if self._h:
            lib().gs_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """Renderer counterpart: one stream, one frame in flight."""

    def __init__(self, scene):
        self.scene = scene
        self._h = C.c_void_p()
        _check(lib().gs_renderer_create(scene._h, C.byref(self._h)))

    def render(self, uniforms, rgba_ptr=0, bgra_ptr=0):
        """Enqueue one frame into device buffers (raw pointers, e.g. tensor.data_ptr())."""
        _check(lib().gs_render(self._h, _p(uniforms), C.c_void_p(rgba_ptr), C.c_void_p(bgra_ptr)))

    def render_host(self, uniforms, want_rgba=True, want_bgra=False):
        h, w = int(uniforms["height"][0]), int(uniforms["width"][0])
        rgba = np.zeros((h, w, 4), np.float32) if want_rgba else None
        bgra = np.zeros((h, w, 4), np.uint8) if want_bgra else None
        _check(lib().gs_render_host(self._h, _p(uniforms), _p(rgba) if want_rgba else None,
                                    _p(bgra) if want_bgra else None))
        return rgba, bgra

    def synchronize(self):
        _check(lib().gs_synchronize(self._h))

    def set_timing(self, enabled):
        _check(lib().gs_set_timing(self._h, C.c_int(int(enabled))))

    def set_frames_in_flight(self, frames):
        _check(lib().gs_set_frames_in_flight(self._h, C.c_int(int(frames))))

    def set_exp_mode(self, mode):
        """3 (default) v_exp_f32 under the guard of render.comp:82 (the reference's decisions, its pixels to rounding noise), 2 libm's expf
        restated in binary64 (the reference's bits), 0 pipeline-defined polynomial, 1 v_exp_f32 unguarded (gs_set_exp_mode)."""
        _check(lib().gs_set_exp_mode(self._h, C.c_int(int(mode))))

    def set_blend_contraction(self, enabled):
        """False (default): render.comp:66,87 as written; True: the three FMA contractions GLSL permits (gs_set_blend_contraction)."""
        _check(lib().gs_set_blend_contraction(self._h, C.c_int(int(bool(enabled)))))

    def set_fast_blend(self, fast):
        """True: both opt-in relaxations (polynomial exp, contractions); False: exp mode 2 as written, bit-identical to the reference text."""
        self.set_exp_mode(0 if fast else 2)
        self.set_blend_contraction(bool(fast))

    def set_graph_mode(self, enabled):
        """Replay frames as captured HIP graphs (gs_set_graph_mode)."""
        _check(lib().gs_set_graph_mode(self._h, C.c_int(int(bool(enabled)))))

    def set_blend_lockstep(self, mode):
        """-1 automatic (measured), 0 off, 1 on: the tile's four waves take every chunk together (gs_set_blend_lockstep)."""
        _check(lib().gs_set_blend_lockstep(self._h, C.c_int(int(mode))))

    def blend_lockstep(self):
        """(the setting the next frame runs with, whether the measurement has settled)."""
        settled = C.c_int(0)
        now = lib().gs_get_blend_lockstep(self._h, C.byref(settled))
        if now < 0:
            _check(now)
        return bool(now), bool(settled.value)

    def set_sort_path(self, mode):
        """0 automatic, 1 global depth order, 2 bin-local (gs_set_sort_path)."""
        _check(lib().gs_set_sort_path(self._h, C.c_int(int(mode))))

    def timing_totals(self, reset=True):
        """(sums of per-pass ms as FrameStats, number of frames summed)."""
        st = FrameStats()
        n = C.c_uint64()
        _check(lib().gs_get_timing_totals(self._h, C.byref(st), C.byref(n), C.c_int(int(reset))))
        return st, n.value

    def frame_intervals(self, reset=True, capacity=8192):
        """ms between the completions of consecutive frames (GPU timestamps), oldest first."""
        out = np.zeros(capacity, np.float32)
        n = C.c_uint64()
        _check(lib().gs_get_frame_intervals(self._h, _p(out), C.c_uint64(capacity), C.byref(n), C.c_int(int(reset))))
        return out[:min(capacity, n.value)].copy()

    def stats(self):
        st = FrameStats()
        _check(lib().gs_get_stats(self._h, C.byref(st)))
        return st

    def poll_stats(self):
        """(stats of the most recently retired frame, frames retired so far) without waiting (gs_poll_stats)."""
        st = FrameStats()
        n = C.c_uint64()
        _check(lib().gs_poll_stats(self._h, C.byref(st), C.byref(n)))
        return st, n.value

    def stage(self, name, uniforms=None):
        """Download a stage buffer of the last frame as a numpy array."""
        code, dt = STAGES[name]
        st = self.stats()
        n, v, d = st.num_gaussians, st.num_visible, min(st.num_instances, st.instance_capacity)
        count = {"tiles": n, "depth": n, "radius": n, "aabb": 4 * n, "conic_opacity": 4 * n, "uv_rg": 4 * n,
                 "b": n, "alpha_cut": n, "depth_order": v, "sorted_tile": d, "sorted_gid": d, "lists_raw": d}.get(name)
        if name in ("ranges", "ranges_raw"):
            w, h = int(uniforms["width"][0]), int(uniforms["height"][0])
            count = 2 * ((w + 15) // 16) * ((h + 15) // 16)
        out = np.zeros(max(int(count), 1), dt)
        _check(lib().gs_debug_download(self._h, C.c_int(code), _p(out), C.c_uint64(int(count) * out.itemsize)))
        return out[:int(count)]

    @property
    def stream(self):
        return lib().gs_renderer_stream(self._h)

    def close(self):
        if self._h:
            lib().gs_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dist:
    """One process per GPU: RCCL communicator + the one scene broadcast (gs_dist_*, include/gs3d_hip.h)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * Dist.ID_BYTES)()
        _check(lib().gs_dist_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, world, device=0):
        h = C.c_void_p()
        buf = (C.c_uint8 * Dist.ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().gs_dist_create(buf, C.c_int(rank), C.c_int(world), C.c_int(device), C.byref(h)))
        self._h = h
        lib().gs_dist_pose_count.restype = C.c_uint64

    @property
    def rank(self):
        return lib().gs_dist_rank(self._h)

    @property
    def world(self):
        return lib().gs_dist_world(self._h)

    def pose_count(self, poses):
        return int(lib().gs_dist_pose_count(self._h, C.c_uint64(poses)))

    def broadcast_scene(self, scene=None, root=0, copy_on_root=False):
        """Root passes its Scene and gets it back (or, with copy_on_root, a new replica like every other rank); the other
        ranks get a new Scene holding the received blob.  A quantised root scene arrives quantised."""
        out = C.c_void_p()
        _check(lib().gs_dist_broadcast_scene_ex(self._h, scene._h if scene is not None else None, C.c_int(root),
                                                C.c_uint(1 if copy_on_root else 0), C.byref(out)))
        if scene is not None and out.value == scene._h.value:
            return scene
        return Scene(out)

    def close(self):
        if self._h:
            lib().gs_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
