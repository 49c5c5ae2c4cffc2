"""CPU-side checks of the product library: it loads, exports the whole C ABI, its host arithmetic
(camera uniforms, PLY ingest + activation) equals the oracle's bit for bit, and it refuses to run the
GPU path without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "gs3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.binding.lib()
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/gs3d_hip.h but not exported"
    assert sorted(pkg.binding.SYMBOLS) == declared


def test_public_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/gs3d_hip.h must compile as strict C99 on its own (what a cgo / JNI / ctypes
    binding generator would feed it to), with no C++ or HIP types in any signature."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "gs3d_hip.h"\nint main(void) { gs_uniforms u; gs_frame_stats s; (void)u; (void)s; return (int)sizeof(gs_camera) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "hdr.o")])
    code = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "gs3d_hip.h")).read(), flags=re.S)
    assert "hipStream_t" not in code and "torch" not in code and "std::" not in code


def test_struct_layouts(pkg):
    assert pkg.binding.UNIFORMS_DT.itemsize == 160  # std140 block, Renderer.h:21-29
    assert pkg.binding.CAMERA_DT.itemsize == 40
    assert ctypes.sizeof(pkg.binding.FrameStats) == 96  # gs_frame_stats: + blend_resolved and a pad word (round 4)


def test_camera_uniforms_match_oracle_bitwise(pkg, oracle):
    rng = np.random.default_rng(0)
    for i in range(200):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        pos = rng.uniform(-3, 3, size=3)
        w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 2200))
        fov = float(rng.uniform(10, 120))
        a = pkg.camera_uniforms(pkg.make_camera(pos, q, fov, 0.1, 1000.0), w, h)
        b = oracle.camera_uniforms(oracle.default_camera(pos, q, fov, 0.1, 1000.0), w, h)
        assert a.tobytes() == b.tobytes(), i


def test_camera_uniforms_rejects_empty_framebuffer(pkg):
    with pytest.raises(pkg.GsError) as e:
        pkg.camera_uniforms(pkg.make_camera(), 0, 10)
    assert e.value.code == -1


def test_activation_matches_oracle_bitwise(pkg, oracle):
    rec = pkg.synth.synth_records(5000, seed=4, kind="S", n_total=1_000_000)
    a = pkg.activate_records(rec)
    b = oracle.activate_records(rec).view(np.float32).reshape(-1, 60)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_ply_roundtrip_and_errors(pkg, oracle, tmp_path):
    rec = pkg.synth.synth_records(777, seed=9, kind="A")
    path = str(tmp_path / "scene.ply")
    pkg.synth.write_ply(path, rec)
    np.testing.assert_array_equal(pkg.read_ply(path), rec)                 # product reader
    np.testing.assert_array_equal(pkg.synth.read_ply_records(path), rec)   # python helper
    ov = oracle.load_ply(path).view(np.float32).reshape(-1, 60)            # oracle reader + activation
    np.testing.assert_array_equal(pkg.activate_records(pkg.read_ply(path)).view(np.uint32), ov.view(np.uint32))
    # GSScene.h:29 "File does not exist: ..."
    with pytest.raises(pkg.GsError) as e:
        pkg.read_ply(str(tmp_path / "missing.ply"))
    assert e.value.code == -2 and "File does not exist" in str(e.value)
    # GSScene.cpp:147 "Could not find end of header"
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\n")
    with pytest.raises(pkg.GsError) as e:
        pkg.read_ply(str(bad))
    assert "end of header" in str(e.value)
    # truncated payload
    trunc = tmp_path / "trunc.ply"
    trunc.write_bytes(open(path, "rb").read()[:-100])
    with pytest.raises(pkg.GsError):
        pkg.read_ply(str(trunc))
    # zero vertices
    empty = str(tmp_path / "empty.ply")
    pkg.synth.write_ply(empty, np.zeros((0, 62), np.float32))
    assert pkg.read_ply(empty).shape == (0, 62)


def test_no_cpu_fallback(pkg):
    """Without a GPU the product must fail loudly (GS_ERR_DEVICE), never compute on the host."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.GsError) as e:
        pkg.Scene.from_records(pkg.synth.synth_records(8))
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under 3dgs.cpp_amd/ or include/ may mention it."""
    for base in ("3dgs.cpp_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "gs_oracle" not in text and "gso_" not in text and "oracle/" not in text, os.path.join(dirpath, f)


def test_synth_scene_is_deterministic_and_sliceable(pkg):
    a = pkg.synth.synth_records(1000, seed=3, kind="S", n_total=4000)
    b = pkg.synth.synth_records(400, seed=3, kind="S", n_total=4000, start=300)
    np.testing.assert_array_equal(a[300:700], b)
    assert not a[:, 3:6].any()  # normals are zero (GSScene.cpp:56-58)
    assert abs(float(a[:, 55:58].mean()) - (-4.5 - np.log(4000 / 1e6) / 3)) < 0.05


def _write_custom_ply(path, names_types, rows):
    """rows: dict name -> array; writes a binary little-endian PLY with the given property order/types."""
    n = len(next(iter(rows.values())))
    dt = np.dtype([(nm, {"float": "<f4", "double": "<f8", "uchar": "u1", "int": "<i4"}[ty]) for nm, ty in names_types])
    data = np.zeros(n, dt)
    for nm, _ in names_types:
        data[nm] = rows[nm]
    header = "ply\nformat binary_little_endian 1.0\ncomment test\nelement vertex %d\n" % n
    header += "".join(f"property {ty} {nm}\n" for nm, ty in names_types) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode())
        f.write(data.tobytes())


def test_ply_name_mapped_layouts(pkg, tmp_path):
    """Beyond the reference (which silently mis-reads anything but the 62-float INRIA order): properties in any
    order, extra / missing-normal properties, lower SH degree -- mapped by name, zero-extended."""
    rec = pkg.synth.synth_records(300, seed=31, kind="A")
    std_names = pkg.synth._PROPS
    cols = {nm: rec[:, k].copy() for k, nm in enumerate(std_names)}
    # (1) shuffled order, no normals, an extra uchar and an extra double property
    order = [nm for nm in std_names if nm not in ("nx", "ny", "nz")]
    rng = np.random.default_rng(5)
    rng.shuffle(order)
    layout = [(nm, "float") for nm in order]
    layout.insert(3, ("label", "uchar"))
    layout.insert(20, ("confidence", "double"))
    cols["label"] = np.arange(300) % 7
    cols["confidence"] = np.linspace(0, 1, 300)
    p1 = str(tmp_path / "shuffled.ply")
    _write_custom_ply(p1, layout, cols)
    np.testing.assert_array_equal(pkg.read_ply(p1), rec)
    # (2) SH degree 1: 9 f_rest values (3 per channel, planar) -> zero-extended to degree 3
    K = 3
    deg1 = [(nm, "float") for nm in std_names if not nm.startswith("f_rest_")]
    deg1 += [(f"f_rest_{i}", "float") for i in range(3 * K)]
    cols1 = dict(cols)
    for c in range(3):
        for j in range(K):
            cols1[f"f_rest_{c * K + j}"] = rec[:, 9 + c * 15 + j]
    p2 = str(tmp_path / "deg1.ply")
    _write_custom_ply(p2, deg1, cols1)
    expect = rec.copy()
    for c in range(3):
        expect[:, 9 + c * 15 + K: 9 + (c + 1) * 15] = 0
    np.testing.assert_array_equal(pkg.read_ply(p2), expect)
    # (3) errors: missing required property, non-float required property, ascii format
    p3 = str(tmp_path / "noopacity.ply")
    _write_custom_ply(p3, [(nm, "float") for nm in std_names if nm != "opacity"], cols)
    with pytest.raises(pkg.GsError) as e:
        pkg.read_ply(p3)
    assert e.value.code == -2 and "opacity" in str(e.value)
    p4 = str(tmp_path / "doublex.ply")
    _write_custom_ply(p4, [(nm, "double" if nm == "x" else "float") for nm in std_names], cols)
    with pytest.raises(pkg.GsError) as e:
        pkg.read_ply(p4)
    assert "32-bit float" in str(e.value)
    p5 = tmp_path / "ascii.ply"
    p5.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    with pytest.raises(pkg.GsError) as e:
        pkg.read_ply(str(p5))
    assert "unsupported PLY format" in str(e.value)


def test_parallel_activation_matches_serial(pkg, oracle):
    """Load-time activation runs on several host threads for large scenes; results must not depend on that."""
    rec = pkg.synth.synth_records(200_000, seed=6, kind="S")
    a = pkg.activate_records(rec)
    b = np.concatenate([pkg.activate_records(rec[i:i + 10_000]) for i in range(0, len(rec), 10_000)])
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    np.testing.assert_array_equal(a[:2000].view(np.uint32), oracle.activate_records(rec[:2000]).view(np.float32).reshape(-1, 60).view(np.uint32))


def test_dense_list_geometry_holds_what_its_workgroups_append(pkg):
    """k_preprocess's workgroup b (256 Gaussians) appends its visible ones to list b % 256 (gs_kernels.h, AttrView::vis).
    A list must hold ALL Gaussians of its workgroups (every one may be visible), be a whole number of level-1 blocks
    (1024 slots), and the level-1 table must have a column for every block of either input (planes or lists).
    Pure host functions of the library: no GPU needed."""
    L = pkg.binding.lib()
    slots_of = getattr(L, "_ZN2gs16vis_region_slotsEj")
    columns_of = getattr(L, "_ZN2gs18bin_level1_columnsEj")
    blocks_of = getattr(L, "_ZN2gs17bin_level1_blocksEj")
    for f in (slots_of, columns_of, blocks_of):
        f.restype = ctypes.c_uint32
        f.argtypes = [ctypes.c_uint32]
    rng = np.random.default_rng(0)
    sizes = [0, 1, 255, 256, 257, 65535, 65536, 65537, 10_000, 1_000_000, 6_000_000, 2**24, 2**24 + 1, 100_000_000, 2**31 - 1]
    sizes += [int(x) for x in rng.integers(1, 2**28, 200)]
    for n in sizes:
        slots = slots_of(n)
        groups = -(-n // 256)
        fullest = -(-groups // 256) * 256  # Gaussians of the workgroups 0, 256, 512, ... (list 0 has the most)
        assert slots >= max(fullest, 1) and slots % 1024 == 0, (n, slots)
        assert slots - fullest < 1024 or n == 0, (n, slots)  # ... and no more than the rounding (an empty scene: one block)
        assert columns_of(n) >= 256 * (slots // 1024) and columns_of(n) >= blocks_of(n), n
        assert 256 * slots * 16 <= 2**40  # 1 TiB: the list index stays far inside 32 bits of slots
        assert 256 * slots < 2**32, n
