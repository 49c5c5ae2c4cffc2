"""Frames-in-flight sweep / A-B of library builds on the GPU box -- no torch, one process, a few seconds:
    [GS3D_HIP_LIB=...] python tools/tune_sweep.py [--quick] [--ref-image /tmp/ref.npy] [--gaussians N --width W --height H]
Every configuration renders the same frame; the first image (or the one in --ref-image, written by an earlier
process) is the reference the others must equal bit for bit, and V / E1 / D must not move.  One line per
configuration: frames/s (median of the timed batches) and the per-pass spans.
(The launch-shape knobs this script once swept -- persistent preprocess grid, blend residency cap, s_setprio, at commit
76662c8 -- lost or were neutral: profiles/r02_knob_sweep.txt.)"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--fif", type=str, default="1,2,3,4,6")
    ap.add_argument("--quick", action="store_true", help="1 and 3 frames in flight, twice (A/B of library builds)")
    ap.add_argument("--sh16", action="store_true")
    ap.add_argument("--scene", choices=["S", "T"], default="S")
    ap.add_argument("--exp-mode", type=int, default=3, help="gs_set_exp_mode: 3 guarded v_exp_f32 (default), 2 libm-exact, 0 polynomial, 1 v_exp_f32")
    ap.add_argument("--contract", type=int, default=0, help="gs_set_blend_contraction")
    ap.add_argument("--warm", type=int, default=20, help="untimed frames ahead of every configuration")
    ap.add_argument("--no-prime", action="store_true", help="skip the initial 3-in-flight run (profiled runs: only the asked configurations launch)")
    ap.add_argument("--json-out", type=str, default="", help="write the last configuration's numbers as one JSON object")
    ap.add_argument("--ref-image", type=str, default="",
                    help="npy file: the first process saves its frame there, later ones must equal it bit for bit")
    args = ap.parse_args()

    pkg = entry.load_package()
    hip = ctypes.CDLL("libamdhip64.so")
    n, w, h = args.gaussians, args.width, args.height
    t0 = time.perf_counter()
    cache = f"/tmp/gs_sweep_scene_{args.scene}{n}.npy"  # a second process of the same call (another library) reuses the scene
    if os.path.exists(cache):
        rec = np.load(cache)
    else:
        rec = pkg.synth.synth_records(n, seed=0, kind=args.scene)
        np.save(cache, rec)
    scene = pkg.Scene.from_records(rec, device=0)
    del rec
    if args.sh16:
        scene.quantize_sh()
    rend = pkg.Renderer(scene)
    rend.set_exp_mode(args.exp_mode)
    rend.set_blend_contraction(bool(args.contract))
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = []
    for _ in range(8):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(w * h * 16)) == 0
        outs.append(p.value)
    print(f"# scene ready in {time.perf_counter() - t0:.1f} s; lib={os.environ.get('GS3D_HIP_LIB', 'default')}", flush=True)

    def download(ptr):
        img = np.zeros((h, w, 4), np.float32)
        assert hip.hipMemcpy(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(img.nbytes), 2) == 0
        return img

    ref_img = np.load(args.ref_image) if args.ref_image and os.path.exists(args.ref_image) else None

    def run(fif):
        nonlocal ref_img
        rend.set_frames_in_flight(fif)
        for i in range(args.warm):
            rend.render(u, outs[i % fif], 0)
        rend.synchronize()
        rend.timing_totals(reset=True)
        fps = []
        for _ in range(args.batches):
            t = time.perf_counter()
            for i in range(args.frames):
                rend.render(u, outs[i % fif], 0)
            rend.synchronize()
            fps.append(args.frames / (time.perf_counter() - t))
        sums, nf = rend.timing_totals(reset=True)
        spans = " ".join(f"{getattr(sums, 'ms_' + k) / max(nf, 1) * 1e3:.0f}" for k in
                         ("preprocess", "prefix_sum", "preprocess_sort", "sort", "render", "total"))
        img = download(outs[0])
        if ref_img is None:
            ref_img = img
            if args.ref_image:
                np.save(args.ref_image, img)
        same = bool(np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)))
        st = rend.stats()
        ls = rend.blend_lockstep() if hasattr(rend, "blend_lockstep") and hasattr(pkg.binding.lib(), "gs_get_blend_lockstep") else (None, None)
        print(f"fif {fif}  fps {np.median(fps):8.1f}  (min {min(fps):8.1f} max {max(fps):8.1f})  bit-equal {same} lockstep {ls[0]}/{'settled' if ls[1] else 'measuring'} "
              f"V {st.num_visible} E1 {st.num_bin_entries} D {st.num_instances} lvl {st.sort_level} path {st.sort_path} bin {st.bin_tiles} maxbin {st.max_bin_entries}  spans us [pre l1cnt l1scat bin blend total] {spans}",
              flush=True)
        if args.json_out:
            import json
            with open(args.json_out, "w") as f:
                json.dump({"driver": "tools/tune_sweep.py", "frames_in_flight": fif, "value": round(float(np.median(fps)), 2),
                           "unit": "frames/s", "spans_us": dict(zip(["preprocess", "prefix_sum", "preprocess_sort", "sort", "render", "total"],
                                                                    [float(x) for x in spans.split()])),
                           "config": {"gaussians": int(st.num_gaussians), "visible": int(st.num_visible),
                                      "instances": int(st.num_instances), "bin_entries": int(st.num_bin_entries),
                                      "width": w, "height": h, "scene": args.scene,
                                      "exp_mode": args.exp_mode, "contract": args.contract}}, f)
                f.write("\n")
        return float(np.median(fps))

    if not args.no_prime:
        run(3)  # reference image + clocks up
    if args.quick:
        for _ in range(2):
            for fif in (1, 3):
                run(fif)
    else:
        for fif in [int(x) for x in args.fif.split(",")]:
            run(fif)


if __name__ == "__main__":
    main()
