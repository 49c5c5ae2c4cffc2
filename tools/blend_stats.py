"""Blend-kernel work statistics from the instrumented build (libgs3d_hip_stats.so, -DGS_BLEND_STATS: `make -C
3dgs.cpp_amd/csrc stats`), per workload, as JSON -- the figures bench.py's `roofline.flops_view` is computed from.

    GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so python tools/blend_stats.py [--out gpurun_out/blend_work.json] B C T E

walked_pairs        (pixel, entry) pairs the REFERENCE's loop walks: every pixel, its tile's list up to and including the
                    entry it breaks at (render.comp:60-85) -- a property of the workload, not of this implementation;
                    SURVEY 8d prices the blend at 22 flop each
contributing_pairs  of those, the ones with alpha >= 1/255 that are accumulated
wave_pairs          (entry, 8x8-pixel wave) pairs this implementation evaluates after its exact quadrant culling
lanes_in_exp        lanes that reach exp() in them
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

WORKLOADS = {"A": (10_000, 256, 256, "A"), "B": (1_000_000, 1920, 1080, "S"), "C": (6_000_000, 1920, 1080, "S"),
             "T": (6_000_000, 1920, 1080, "T"), "E": (6_000_000, 3840, 2160, "S")}


def workload_key(n, w, h, kind):
    return f"{kind}({n})@{w}x{h}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["B"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    pkg = entry.load_package()
    L = pkg.binding.lib()
    if not hasattr(L, "gs_debug_blend_stats"):
        raise SystemExit("needs the instrumented build: GS3D_HIP_LIB=3dgs.cpp_amd/libgs3d_hip_stats.so")
    result = {}
    for name in args.names:
        n, w, h, kind = WORKLOADS[name]
        rec = pkg.synth.synth_records(n, seed=0, kind=kind)
        scene = pkg.Scene.from_records(rec)
        del rec
        rend = pkg.Renderer(scene)
        u = pkg.camera_uniforms(pkg.make_camera(), w, h)
        out = (ctypes.c_ulonglong * 12)()
        for _ in range(3):  # the first frames may be re-run at a larger sort level: keep the last, clean one
            rend.render_host(u, want_rgba=True)
            L.gs_debug_blend_stats(out, 1)
        st = rend.stats()
        v = [int(x) for x in out]
        result[workload_key(n, w, h, kind)] = {
            "gaussians": int(st.num_gaussians), "visible": int(st.num_visible), "instances": int(st.num_instances),
            "walked_pairs": v[7], "contributing_pairs": v[8], "wave_pairs": v[2], "wave_pairs_reaching_exp": v[4],
            "lanes_alive_in_wave_pairs": v[3], "lanes_in_exp": v[5], "wave_chunks": v[0], "entries_staged": v[6],
            "flop_per_walked_pair": 22}
        print(name, json.dumps(result[workload_key(n, w, h, kind)]), flush=True)
        rend.close()
        scene.close()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
