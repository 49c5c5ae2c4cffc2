#!/usr/bin/env python3
"""Round-3 survey of the blend's modes on the GPU: parity of each (exp, contraction) pair against the oracle's matching
reading and the reference text, and what each costs (frames/s at config B with 3 frames in flight, serial blend ms)."""
import json
import os
import sys
import time

import numpy as np
import torch  # before the package: both must bind the same libamdhip64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg, oracle, gsref = entry.load_package(), entry.load_oracle(), entry.load_ref()


def needles():
    n2, rng = 6000, np.random.default_rng(31)
    rec = pkg.synth.synth_records(n2, seed=31, kind="A")
    rec[:, 55] = rng.uniform(-1.5, 0.0, n2)
    rec[:, 56:58] = rng.uniform(-9.0, -6.0, (n2, 2))
    rec[:, 58:62] = rng.normal(size=(n2, 4))
    rec[:, 54] = rng.uniform(0.0, 4.0, n2)
    return rec


def parity(label, rec, w, h):
    verts = oracle.activate_records(rec)
    u_ref = oracle.camera_uniforms(oracle.default_camera(), w, h)
    so = oracle.stages(verts, u_ref)
    with oracle.reference_reading():
        strict = oracle.render(so["attr"], so["boundaries"], so["sorted_payload"], w, h)
    rimg = gsref.render(so["attr"], so["boundaries"], so["sorted_payload"], w, h) if gsref.available() else None
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    out = {}
    for exp_mode in (0, 1, 2):
        for contract in (True, False):
            rend.set_exp_mode(exp_mode)
            rend.set_blend_contraction(contract)
            img, _ = rend.render_host(u)
            key = f"exp{exp_mode}{'c' if contract else 'u'}"
            out[key] = {"vs_oracle_default": float(np.abs(img - so["image"]).max()),
                        "vs_oracle_strict": float(np.abs(img - strict).max()),
                        "bits_differ_vs_strict": int((img.view(np.uint32) != strict.view(np.uint32)).sum()),
                        "vs_ref_text": float(np.abs(img - rimg).max()) if rimg is not None else None,
                        "px_gt_1e-4_vs_ref": int((np.abs(img - rimg).max(axis=2) > 1e-4).sum()) if rimg is not None else None}
    print(label, json.dumps(out, indent=1), flush=True)
    rend.close()
    scene.close()


def timing():
    n, w, h = 1_000_000, 1920, 1080
    rec = pkg.synth.synth_records(n, seed=0, kind="S")
    scene = pkg.Scene.from_records(rec, device=0)
    rend = pkg.Renderer(scene)
    u = pkg.camera_uniforms(pkg.make_camera(), w, h)
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    res = {}
    for rep in range(2):
        for exp_mode in (0, 1, 2):
            for contract in (True, False):
                rend.set_exp_mode(exp_mode)
                rend.set_blend_contraction(contract)
                rend.set_frames_in_flight(3)
                for i in range(60):
                    rend.render(u, outs[i % 3].data_ptr(), 0)
                rend.synchronize()
                t0 = time.perf_counter()
                for i in range(600):
                    rend.render(u, outs[i % 3].data_ptr(), 0)
                rend.synchronize()
                fps = 600 / (time.perf_counter() - t0)
                rend.set_frames_in_flight(1)
                rend.timing_totals(reset=True)
                for i in range(100):
                    rend.render(u, outs[0].data_ptr(), 0)
                rend.synchronize()
                s, f = rend.timing_totals(reset=True)
                res.setdefault(f"exp{exp_mode}{'c' if contract else 'u'}", []).append(
                    {"fps": round(fps, 1), "blend_ms_serial": round(s.ms_render / f, 4), "total_ms_serial": round(s.ms_total / f, 4)})
    print("timing", json.dumps(res, indent=1), flush=True)


if __name__ == "__main__":
    if "--timing-only" in sys.argv:
        timing()
        sys.exit(0)
    parity("config A", pkg.synth.synth_records(10000, seed=0, kind="A"), 256, 256)
    parity("needles", needles(), 640, 360)
    timing()
