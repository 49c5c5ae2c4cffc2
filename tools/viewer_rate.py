#!/usr/bin/env python3
"""Frames/s of the reference's UNCHANGED viewer (apps/viewer/main.cpp built against lib3dgs_cpp.so: viewer_ref) running headless on
config B's scene -- the consumer the drop-in library is for.  GS_FRAMES_IN_FLIGHT=1 is the reference's own
mode (VulkanContext.h:6); the default keeps three frames queued.
    python tools/viewer_rate.py [--frames 5000] [--gaussians 1000000]"""
import argparse, os, subprocess, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5000)
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
pkg = entry.load_package()
exe = os.path.join(entry.PKG_DIR, "viewer_ref")
with tempfile.TemporaryDirectory() as tmp:
    ply = os.path.join(tmp, "scene.ply")
    pkg.synth.write_ply(ply, pkg.synth.synth_records(a.gaussians, seed=0, kind="S"))
    for fif in ("1", "3", "1", "3"):
        rates = {}
        for frames in (200, a.frames + 200):  # two runs: the difference cancels process start, PLY load and clock ramp
            env = dict(os.environ, GS_FRAMES=str(frames), GS_FRAMES_IN_FLIGHT=fif)
            t = time.perf_counter()
            out = subprocess.run([exe, "--no-gui", "--width", str(a.width), "--height", str(a.height), ply], env=env, capture_output=True, text=True, timeout=600)
            rates[frames] = time.perf_counter() - t
            assert out.returncode == 0, out.stderr[-2000:]
        dt = rates[a.frames + 200] - rates[200]
        fps_lines = [l for l in out.stderr.splitlines() if "FPS" in l]
        print(f"viewer_ref --no-gui, {a.gaussians} Gaussians {a.width}x{a.height}, GS_FRAMES_IN_FLIGHT={fif}: {a.frames} frames in {dt:.3f} s = {a.frames / dt:.1f} frames/s"
              f"   (the viewer's own once-a-second counter: {fps_lines[-1].strip() if fps_lines else 'n/a'})", flush=True)
