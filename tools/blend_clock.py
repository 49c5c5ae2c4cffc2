"""The shader clock k_blend actually runs at (make -C 3dgs.cpp_amd/csrc variant TAG=clk DEFS=-DGS_BLEND_CLOCK): s_memtime cycles over
s_memrealtime ticks around the SHIPPED pair loops, summed over the kernel's waves.  With it the VALU issue roof of THIS kernel on THIS
chip is known: 1024 SIMDs x clock / 2 wave64-instructions per second.
    GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_clk.so python tools/blend_clock.py [B C T E] [--exp-mode 3]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

WORKLOADS = {"B": (1_000_000, 1920, 1080, "S"), "C": (6_000_000, 1920, 1080, "S"), "T": (6_000_000, 1920, 1080, "T"), "E": (6_000_000, 3840, 2160, "S")}
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="*", default=["B"])
ap.add_argument("--exp-mode", type=int, default=3)
ap.add_argument("--fif", type=int, default=1)
args = ap.parse_args()
pkg = entry.load_package()
L = pkg.binding.lib()
if not hasattr(L, "gs_debug_blend_stats"):
    raise SystemExit("needs GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_clk.so")
hip = ctypes.CDLL("libamdhip64.so")
for name in args.names:
    n, w, h, kind = WORKLOADS[name]
    scene = pkg.Scene.from_records(pkg.synth.synth_records(n, seed=0, kind=kind))
    rend = pkg.Renderer(scene)
    rend.set_exp_mode(args.exp_mode)
    rend.set_blend_lockstep(0 if kind == "S" else 1)
    rend.set_frames_in_flight(args.fif)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    ptrs = []
    for _ in range(args.fif):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(w * h * 16)) == 0
        ptrs.append(p.value)
    for i in range(300):  # warm: level settled, clocks up
        rend.render(u, ptrs[i % args.fif], 0)
    rend.synchronize()
    frames = 200
    rend.timing_totals(reset=True)
    for i in range(frames):
        rend.render(u, ptrs[i % args.fif], 0)
    rend.synchronize()
    sums, nfr = rend.timing_totals(reset=True)
    clk = (ctypes.c_ulonglong * 2)()
    assert L.gs_debug_blend_clock(clk, ctypes.c_uint(((w + 15) // 16) * ((h + 15) // 16) * 4)) == 0  # the last launch's waves
    ghz = clk[0] / max(clk[1], 1) * 0.1
    print(json.dumps({"workload": name, "exp_mode": args.exp_mode, "frames_in_flight": args.fif, "shader_clock_GHz_in_k_blend": round(ghz, 3),
                      "blend_span_us": round(1e3 * sums.ms_render / max(nfr, 1), 1),
                      "valu_issue_roof_G_wave_insts_per_s": round(1024 * ghz / 2, 1)}), flush=True)
    rend.close()
    scene.close()
