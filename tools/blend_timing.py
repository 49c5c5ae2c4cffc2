"""Where a blend wave's cycles go (instrumented build: make variant TAG=tm1 DEFS=-DGS_BLEND_TIMING): per chunk, loop top .. keep-ballot
(the wait for the prefetched records + classification) and ballot .. end of the pair loop (staging + pairs).
    GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_tm1.so python tools/blend_timing.py B T"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
W = {"B": (1_000_000, 1920, 1080, "S"), "C": (6_000_000, 1920, 1080, "S"), "T": (6_000_000, 1920, 1080, "T"), "E": (6_000_000, 3840, 2160, "S")}
pkg = entry.load_package()
L = pkg.binding.lib()
for name in sys.argv[1:] or ["B"]:
    n, w, h, kind = W[name]
    scene = pkg.Scene.from_records(pkg.synth.synth_records(n, seed=0, kind=kind))
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    out = (ctypes.c_ulonglong * 12)()
    for _ in range(4):
        rend.render_host(u)
        L.gs_debug_blend_stats(out, 1)
    st = rend.stats()
    v = [int(x) for x in out]
    print(f"{name}: lib {os.environ.get('GS3D_HIP_LIB','default')}  blend {st.ms_render*1e3:.0f} us  chunks with a kept entry {v[11]}  "
          f"top..ballot {v[9]/1e6:.1f} Mcycles  ballot..end-of-pairs {v[10]/1e6:.1f} Mcycles  per kept chunk: {v[10]/max(v[11],1):.0f} cycles", flush=True)
    rend.close(); scene.close()
