#!/usr/bin/env python3
"""Round 4: the guarded blend (exp mode 3) beside the other modes -- frames/s with 3 frames in flight and one at a time, serial
blend ms, quadrants re-rendered, distance from the exact mode's frame (= the reference text's, bit for bit).
usage: r04_guard.py [B|C|T|E ...]"""
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
        "E": (6_000_000, 3840, 2160, "S"), "A": (10_000, 256, 256, "A")}
MODES = [("exact (2)", 2, False), ("guarded (3)", 3, False), ("v_exp_f32 (1)", 1, False), ("poly (0)", 0, False),
         ("poly+contract (0c)", 0, True), ("v_exp_f32+contract (1c)", 1, True)]


def run(name):
    n, w, h, kind = WORK[name]
    rec = pkg.synth.synth_records(n, seed=0, kind=kind)
    scene = pkg.Scene.from_records(rec, device=0)
    del rec
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    rend.set_exp_mode(2)
    exact, _ = rend.render_host(u)
    rend.render_host(u)
    res = {}
    frames = 600 if name in ("A", "B") else 200
    for rep in range(3):
        for label, exp_mode, contract in MODES:
            rend.set_exp_mode(exp_mode)
            rend.set_blend_contraction(contract)
            img, _ = rend.render_host(u)
            st = rend.stats()
            d = np.abs(img[..., :3].astype(np.float64) - exact[..., :3]).max(axis=2)
            rend.set_frames_in_flight(3)
            for i in range(frames // 5):
                rend.render(u, outs[i % 3].data_ptr(), 0)
            rend.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                rend.render(u, outs[i % 3].data_ptr(), 0)
            rend.synchronize()
            fps = frames / (time.perf_counter() - t0)
            rend.set_frames_in_flight(1)
            rend.timing_totals(reset=True)
            t0 = time.perf_counter()
            for i in range(frames // 4):
                rend.render(u, outs[0].data_ptr(), 0)
            rend.synchronize()
            fps1 = (frames // 4) / (time.perf_counter() - t0)
            tot, nf = rend.timing_totals(reset=True)
            r = res.setdefault(label, {"fps": [], "fps_one": [], "blend_ms": []})
            r["fps"].append(round(fps, 1))
            r["fps_one"].append(round(fps1, 1))
            r["blend_ms"].append(round(tot.ms_render / max(nf, 1), 4))
            r["max_abs_vs_exact"] = float(d.max())
            r["px_gt_1e-5"] = int((d > 1e-5).sum())
            r["px_gt_1e-4"] = int((d > 1e-4).sum())
            r["blend_redo"] = int(st.blend_redo)
            r["blend_resolved"] = int(st.blend_resolved)
    quads = ((w + 7) // 8) * ((h + 7) // 8)
    print(json.dumps({"workload": name, "N": n, "res": [w, h], "quadrants": quads, "modes": res}), flush=True)
    rend.close()
    scene.close()


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["B"]):
        run(name)
