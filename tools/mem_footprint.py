#!/usr/bin/env python3
"""Device memory a renderer takes per frame set (hipMemGetInfo deltas): scene, renderer with one set, each further set of frames in flight.
    python tools/mem_footprint.py B E      (the level-1 candidates sized on their own)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
W = {"B": (1_000_000, 1920, 1080, "S"), "C": (6_000_000, 1920, 1080, "S"), "T": (6_000_000, 1920, 1080, "T"), "E": (6_000_000, 3840, 2160, "S")}
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value / 1e6
pkg = entry.load_package()
for name in sys.argv[1:] or ["B"]:
    n, w, h, kind = W[name]
    rec = pkg.synth.synth_records(n, seed=0, kind=kind)
    m0 = free_mb()
    scene = pkg.Scene.from_records(rec)
    del rec
    m1 = free_mb()
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    rend.render_host(u)
    st = rend.stats()
    m2 = free_mb()
    rend.set_frames_in_flight(3)
    for _ in range(6):
        rend.render_host(u)
    m3 = free_mb()
    print(f"{name}: N {n} V {st.num_visible} E1 {st.num_bin_entries} D {st.num_instances} instance capacity {st.instance_capacity}: scene {m0 - m1:.0f} MB, "
          f"renderer + first frame set {m1 - m2:.0f} MB (incl. the {w}x{h} RGBA32F target of render_host while it lives), two more sets {m2 - m3:.0f} MB "
          f"= {(m2 - m3) / 2:.0f} MB per set; level-1 candidates need {st.num_bin_entries * 12 / 1e6:.1f} MB, lists {st.num_instances * 4 / 1e6:.1f} MB", flush=True)
    rend.close(); scene.close()
