#!/usr/bin/env python3
"""Round 4, verdict item 5: the scene read in spatial (Morton) order by the per-frame kernels (GS_SPATIAL_MIN) against the
scene's own order -- same library, same box, interleaved: frames/s (3 in flight and one at a time), serial per-pass ms, and
the frame compared bit for bit (exact mode) between the two orders.
usage: r04_spatial_ab.py [B|C|T|E ...]"""
import json
import os
import sys
import time

import numpy as np
import torch  # before the package: both must bind the same libamdhip64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
WORK = {"B": (1_000_000, 1920, 1080, "S"), "C": (6_000_000, 1920, 1080, "S"), "T": (6_000_000, 1920, 1080, "T"),
        "E": (6_000_000, 3840, 2160, "S")}


def run(name):
    n, w, h, kind = WORK[name]
    rec = pkg.synth.synth_records(n, seed=0, kind=kind)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    scenes = {}
    BIG = str(1 << 40)
    variants = (("scene order", BIG, None), ("spatial order", "0", None))
    if os.environ.get("GS_AB_DENSE"):  # small scenes: the dense lists forced on / off as well
        variants = (("planes, scene order", BIG, BIG), ("lists, scene order", BIG, "0"), ("lists, spatial order", "0", "0"))
    rends = {}
    for label, spatial_min, dense_min in variants:
        os.environ["GS_SPATIAL_MIN"] = spatial_min
        scenes[label] = pkg.Scene.from_records(rec, device=0)
        if dense_min is not None:
            os.environ["GS_L1_DENSE_MIN"] = dense_min
        rends[label] = pkg.Renderer(scenes[label])
    os.environ.pop("GS_SPATIAL_MIN", None)
    os.environ.pop("GS_L1_DENSE_MIN", None)
    del rec
    images = {}
    for k, r in rends.items():
        r.set_exp_mode(2)
        images[k] = r.render_host(u)[0]
        r.set_exp_mode(3)
    res = {k: {"fps": [], "fps_one": [], "passes_serial_ms": None} for k in rends}
    frames = 400 if name == "B" else 150
    for rep in range(3):
        for k, r in rends.items():
            r.set_frames_in_flight(3)
            for i in range(frames // 4):
                r.render(u, outs[i % 3].data_ptr(), 0)
            r.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                r.render(u, outs[i % 3].data_ptr(), 0)
            r.synchronize()
            res[k]["fps"].append(round(frames / (time.perf_counter() - t0), 1))
            r.set_frames_in_flight(1)
            r.timing_totals(reset=True)
            t0 = time.perf_counter()
            for i in range(frames // 3):
                r.render(u, outs[0].data_ptr(), 0)
            r.synchronize()
            res[k]["fps_one"].append(round((frames // 3) / (time.perf_counter() - t0), 1))
            tot, nf = r.timing_totals(reset=True)
            res[k]["passes_serial_ms"] = {p: round(getattr(tot, "ms_" + p) / nf, 4) for p in
                                          ("preprocess", "prefix_sum", "preprocess_sort", "sort", "render", "total")}
    first = next(iter(images.values()))
    same = all(bool(np.array_equal(first.view(np.uint32), im.view(np.uint32))) for im in images.values())
    print(json.dumps({"workload": name, "N": n, "res": [w, h], "exact_frames_bit_identical": same, "orders": res}), flush=True)
    for r in rends.values():
        r.close()
    for s in scenes.values():
        s.close()


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["C"]):
        run(name)
