"""Sweep of the launch-shape knobs (gs_set_tuning) and of the frames in flight on the GPU box -- no torch, one process:
    python tools/tune_sweep.py [--gaussians N --width W --height H] [--quick]  > gpurun_out/sweep.txt
Every configuration renders the same frame; the first one's image is the reference the others must equal bit for bit
(the knobs change how the chip is shared, never a result).  Prints one line per configuration: frames/s (median of
the timed batches) and the serial frame time."""
import argparse
import ctypes
import itertools
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
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--fif", type=str, default="1,2,3,4,6")
    ap.add_argument("--pre", type=str, default="0,256,512,1024,2048")
    ap.add_argument("--pad", type=str, default="0,11008,14848")   # 8 (no cap) / 7 / 6 blend workgroups per CU
    ap.add_argument("--prio", type=str, default="0,0x333,0x300,0x033")
    ap.add_argument("--cross", action="store_true", help="full cross product instead of one knob at a time")
    ap.add_argument("--quick", action="store_true", help="only the default knobs at 1 and 3 frames in flight (A/B of library builds)")
    ap.add_argument("--ref-image", type=str, default="", help="npy file: the first process saves its frame there, later ones must equal it bit for bit")
    ap.add_argument("--fif-only", action="store_true", help="only the frames-in-flight sweep (e.g. under GPU_MAX_HW_QUEUES=8)")
    args = ap.parse_args()

    pkg = entry.load_package()
    hip = ctypes.CDLL("libamdhip64.so")
    n, w, h = args.gaussians, args.width, args.height
    t0 = time.perf_counter()
    cache = f"/tmp/gs_sweep_scene_{n}.npy"  # a second process of the same call (another environment) reuses the scene
    if os.path.exists(cache):
        rec = np.load(cache)
    else:
        rec = pkg.synth.synth_records(n, seed=0, kind="S")
        np.save(cache, rec)
    scene = pkg.Scene.from_records(rec, device=0)
    del rec
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = []
    for _ in range(8):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(w * h * 16)) == 0
        outs.append(p.value)
    print(f"# scene ready in {time.perf_counter() - t0:.1f} s; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} "
          f"lib={os.environ.get('GS3D_HIP_LIB', 'default')}", flush=True)

    def download(ptr):
        img = np.zeros((h, w, 4), np.float32)
        assert hip.hipMemcpy(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(img.nbytes), 2) == 0
        return img

    ref_img = np.load(args.ref_image) if args.ref_image and os.path.exists(args.ref_image) else None

    def run(fif, pre, pad, prio):
        nonlocal ref_img
        rend.set_tuning("pre_wgs", pre)
        rend.set_tuning("blend_lds_pad", pad)
        rend.set_tuning("prio", prio)
        rend.set_frames_in_flight(fif)
        for i in range(20):
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
        print(f"fif {fif} pre_wgs {pre:5d} pad {pad:6d} prio {prio:#05x}  fps {np.median(fps):8.1f}  (min {min(fps):8.1f} max {max(fps):8.1f})  "
              f"bit-equal {same} V {st.num_visible} E1 {st.num_bin_entries} D {st.num_instances}  spans us [pre l1cnt l1scat bin blend total] {spans}", flush=True)
        return float(np.median(fps))

    fifs = [int(x) for x in args.fif.split(",")]
    pres = [int(x) for x in args.pre.split(",")]
    pads = [int(x) for x in args.pad.split(",")]
    prios = [int(x, 0) for x in args.prio.split(",")]
    run(3, 0, 0, 0)  # reference image + clocks up
    results = {}
    if args.quick:
        for rep in range(2):
            for fif in (1, 3):
                results[(fif, 0, 0, 0)] = run(fif, 0, 0, 0)
    elif args.cross:
        for fif, pre, pad, prio in itertools.product(fifs, pres, pads, prios):
            results[(fif, pre, pad, prio)] = run(fif, pre, pad, prio)
    else:
        for fif in fifs:                       # frames in flight alone
            results[(fif, 0, 0, 0)] = run(fif, 0, 0, 0)
        best_fif = max(fifs, key=lambda f: results[(f, 0, 0, 0)])
        for f in ([] if args.fif_only else sorted({3, best_fif})):        # one knob at a time at the default and at the best depth
            for pre in pres[1:]:
                results[(f, pre, 0, 0)] = run(f, pre, 0, 0)
            for pad in pads[1:]:
                results[(f, 0, pad, 0)] = run(f, 0, pad, 0)
            for prio in prios[1:]:
                results[(f, 0, 0, prio)] = run(f, 0, 0, prio)
            # the best of each knob together
            bp = max(pres, key=lambda x: results.get((f, x, 0, 0), 0))
            bd = max(pads, key=lambda x: results.get((f, 0, x, 0), 0))
            br = max(prios, key=lambda x: results.get((f, 0, 0, x), 0))
            if (bp, bd, br) != (0, 0, 0):
                results[(f, bp, bd, br)] = run(f, bp, bd, br)
                for f2 in fifs:
                    if (f2, bp, bd, br) not in results:
                        results[(f2, bp, bd, br)] = run(f2, bp, bd, br)
    best = max(results, key=results.get)
    print(f"# best: fif {best[0]} pre_wgs {best[1]} pad {best[2]} prio {best[3]:#05x} -> {results[best]:.1f} frames/s; "
          f"default (3, 0, 0, 0) -> {results.get((3, 0, 0, 0), float('nan')):.1f}", flush=True)


if __name__ == "__main__":
    main()
