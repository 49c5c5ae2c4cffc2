#!/usr/bin/env python3
"""Where does the once-per-run stall of a long frame loop come from?  

Renders `--frames` frames of config B through the C ABI with per-call host clocks around gs_render and reports every call
that held the host for more than `--threshold-ms`, the frame index it happened at, the GPU-side completion intervals around
it, and sustained vs median throughput.  No torch unless --torch (the bench harness imports it: one of the suspects).

    python tools/stall_probe.py --frames 8000 --fif 3 [--timing 0] [--graph] [--torch] [--n 1000000]
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8000)
    ap.add_argument("--fif", type=int, default=3)
    ap.add_argument("--timing", type=int, default=1)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--torch", action="store_true")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--threshold-ms", type=float, default=2.0)
    ap.add_argument("--sync-every", type=int, default=0, help="gs_synchronize every k frames (the bench's batches)")
    ap.add_argument("--label", default="")
    ap.add_argument("--legacy-pointers", action="store_true",
                    help="pass numpy pointers the way the binding did up to round 4 (ndarray.ctypes.data_as: two objects of cyclic garbage per call)")
    args = ap.parse_args()

    if args.torch:
        import torch
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
    pkg = entry.load_package()
    if args.legacy_pointers:
        pkg.binding._p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    gc_events = []  # (generation, start ns, ms, objects collected) of every garbage collection of the interpreter

    def gc_cb(phase, info):
        if phase == "start":
            gc_cb.t = time.perf_counter_ns()
        else:
            gc_events.append((info["generation"], gc_cb.t, (time.perf_counter_ns() - gc_cb.t) * 1e-6, info["collected"]))
    gc.callbacks.append(gc_cb)
    hip = ctypes.CDLL("libamdhip64.so")
    w, h = args.width, args.height
    rec = pkg.synth.synth_records(args.n, seed=0, kind="S")
    scene = pkg.Scene.from_records(rec, device=0)
    del rec
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(args.fif)
    rend.set_timing(bool(args.timing))
    if args.graph:
        rend.set_graph_mode(True)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = []
    for _ in range(args.fif):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(w * h * 16)) == 0
        outs.append(p.value)
    for i in range(20):
        rend.render(u, outs[i % args.fif])
    rend.synchronize()
    rend.frame_intervals(reset=True)
    n = args.frames
    t_in = np.zeros(n, np.int64)
    t_out = np.zeros(n, np.int64)
    clock = time.perf_counter_ns
    t0 = clock()
    for i in range(n):
        t_in[i] = clock()
        rend.render(u, outs[i % args.fif])
        t_out[i] = clock()
        if args.sync_every and (i + 1) % args.sync_every == 0:
            rend.synchronize()
    rend.synchronize()
    t1 = clock()
    gc.callbacks.remove(gc_cb)
    call_ms = (t_out - t_in) * 1e-6
    gap_ms = np.concatenate([[0.0], (t_in[1:] - t_out[:-1]) * 1e-6])  # host time BETWEEN calls (the harness itself)
    total_s = (t1 - t0) * 1e-9
    thr = args.threshold_ms
    slow = [{"frame": int(i), "at_s": round(float((t_in[i] - t0) * 1e-9), 4), "call_ms": round(float(call_ms[i]), 3)}
            for i in np.nonzero(call_ms > thr)[0]]
    slow_gaps = [{"before_frame": int(i), "ms": round(float(gap_ms[i]), 3)} for i in np.nonzero(gap_ms > thr)[0]]
    iv = rend.frame_intervals(reset=True)  # the last <= 8192 completion intervals (GPU timestamps)
    big_iv = [{"index_from_end": int(len(iv) - i), "ms": round(float(iv[i]), 3)} for i in np.nonzero(iv > thr)[0]]
    # windows of 100 frames: sustained vs median
    win = 100
    per = np.array([(t_out[min(k + win, n) - 1] - t_in[k]) * 1e-9 / (min(k + win, n) - k) for k in range(0, n, win)])
    out = {"label": args.label, "frames": n, "fif": args.fif, "timing": args.timing, "graph": args.graph, "torch": args.torch,
           "n": args.n, "res": [w, h], "sync_every": args.sync_every,
           "total_s": round(total_s, 4), "sustained_fps": round(n / total_s, 1),
           "median_window_fps": round(1.0 / float(np.median(per)), 1),
           "call_ms": {"p50": round(float(np.median(call_ms)), 4), "p99": round(float(np.percentile(call_ms, 99)), 4),
                       "max": round(float(call_ms.max()), 3)},
           "slow_calls": slow, "slow_harness_gaps": slow_gaps, "gpu_completion_gaps": big_iv,
           "lost_ms_in_slow_calls": round(float(call_ms[call_ms > thr].sum()), 2),
           "python_gc": {"tracked_objects": len(gc.get_objects()), "legacy_pointers": args.legacy_pointers,
                         "collections_in_loop": [{"generation": g, "at_s": round((t - t0) * 1e-9, 4), "ms": round(ms, 3), "collected": c,
                                                  "during_frame": int(np.searchsorted(t_in, t, side="right") - 1)}
                                                 for g, t, ms, c in gc_events if t0 <= t <= t1 and (g == 2 or ms > 0.5)],
                         "count_by_generation": [sum(1 for e in gc_events if e[0] == g and t0 <= e[1] <= t1) for g in range(3)]}}
    print(json.dumps(out), flush=True)
    rend.close()
    scene.close()


if __name__ == "__main__":
    main()
