#!/usr/bin/env python3
"""bench.py -- frames/s of the splat hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: preprocess -> level-1 binning (count, scan, scatter) -> per-bin depth order + tile
lists -> blend, over the synthetic scene S(1e6) at 1920x1080 (BASELINE configs[1]); the scene is resident in
HBM before the timed region and the RGBA32F frame stays in HBM.  The blend runs in the library's DEFAULT mode -- exp mode 3:
the hardware's v_exp_f32 with the reference's decisions (render.comp:78 decided on the alpha cut, render.comp:82 guarded,
ambiguous breaks resolved with the reference's arithmetic): within rounding noise of the reference's render.comp compiled for
the CPU on any scene, BASELINE.json's bar being 1e-4.  The bit-identical mode (exp mode 2) and the opt-in fast modes are timed
beside it as diagnostics (`frames_per_s_exact`, `frames_per_s_fast_blend`, `frames_per_s_hw_exp`) and the measured distance of
each from the reference text on the benched frame is in `parity`.  `--ply FILE` (or GS_SCENE=FILE) benches a trained scene
instead of the synthetic one (BASELINE configs[2]).  After the headline measurement the line also carries `other_configs`:
configs[4] (S(6e6) at 3840x2160) and the trained-like T(6e6) at full size, time-boxed.  With N > 1 the scene blob is
broadcast once over RCCL/xGMI and every rank renders its own camera pose (configs[3]): no per-frame collective, weak
scaling, value = N*K frames / max-over-ranks time; the line's `rccl` block proves what the collective saw.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant pass
and `cpu_baseline` (the oracle -- CPU restatement of the reference shaders -- on the host cores).
"""
import argparse
import ctypes
import gc
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
# FP32 VALU issue roof (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, a wave64 VALU instruction issues over 2 cycles, max
# clock 2.4 GHz) in wave64 instructions per second; and the rate tools/ubench/valu_rate.hip sustains on this chip under
# dense VALU load (1.09 ns per instruction and SIMD, i.e. ~1.85 GHz): a measurement, quoted beside the spec, never as it
VALU_PEAK = 1024 * 2.4e9 / 2
VALU_SUSTAINED = 1024 / 1.09e-9
MIN_WARM_SECONDS = 0.3   # untimed K-step batches run at least this long (the clocks of the driver's fresh box ramp for ~50 ms: BENCH_r05's first batches)
MIN_TIMED_SECONDS = 0.5  # a timed region shorter than this is repeated and the median batch reported
# the blend's modes (gs_set_exp_mode, gs_set_blend_contraction).  "default" is the library's: the guarded v_exp_f32
MODES = {"default": (3, False), "exact": (2, False), "fast": (0, True), "hw_exp": (1, True)}
BLEND_TEXT = {"default": "library default (exp mode 3): v_exp_f32 with the reference's decisions -- render.comp:78 decided on the alpha cut, "
                         "render.comp:82 guarded, ambiguous breaks resolved with the reference's arithmetic; render.comp:66,87 uncontracted; "
                         "within rounding noise of the reference text compiled for the CPU (BASELINE's bar: 1e-4)",
              "exact": "bit-identical (exp mode 2): render.comp:66,87 uncontracted, exp = libm's expf restated in binary64",
              "fast": "opt-in fast: polynomial exp + the three FMA contractions GLSL permits",
              "hw_exp": "opt-in fastest: v_exp_f32 + contractions, unguarded"}


def sig(x, digits=7):
    """x to `digits` SIGNIFICANT digits.  Rates and times of the line are rounded this way, never to fixed decimals: round 5's
    `ms_per_step` had four decimals, and on a 21 000 frames/s workload that quantum alone (1e-3 of the value) broke the
    `value == 1e3 / ms_per_step` identity the contract tests check."""
    return float(f"{x:.{digits}g}") if x is not None and math.isfinite(x) else x


def headline_numbers(world, steps, elapsed_s):
    """`value` (whole-job frames/s: every rank's K frames over the max-over-ranks wall time of the median batch) and
    `ms_per_step`, consistent with each other to 1e-6 relative at any rate (tests/test_bench_contract.py pushes 50 000 and
    200 000 frames/s through it)."""
    return {"value": sig(world * steps / elapsed_s), "ms_per_step": sig(1e3 * elapsed_s / steps)}


def workload_key(n, w, h, kind):
    """Key of a workload in the committed per-workload files (profiles/rNN_pmc_hbm_traffic.json, rNN_blend_work.json)."""
    return f"{kind}({n})@{w}x{h}"


def workload_name(n, w, h, world, kind="S", ply=None):
    """Which BASELINE.json config the arguments are, or what they are when they are none of them."""
    if ply:
        return (f"PLY {os.path.basename(ply)}: {n} Gaussians, {w}x{h}, degree-3 SH, one camera pose per GPU (a trained scene given at "
                "run time: BASELINE configs[2] when it is the Mip-NeRF360 garden file)")
    base = f"{kind}({n}) synthetic Gaussians, {w}x{h}, degree-3 SH, one camera pose per GPU"
    if kind == "T":
        return base + (" (trained-scene statistics: needles and discs, clustered positions, bimodal opacity -- synth.py; "
                       + ("stand-in for BASELINE configs[2], beside S(6000000)" if (n, w, h) == (6_000_000, 1920, 1080)
                          else "not a BASELINE config") + ")")
    if (n, w, h) == (1_000_000, 1920, 1080):
        return base + (" (BASELINE configs[1])" if world == 1 else f" (BASELINE configs[3]: configs[1]'s scene, {world} poses)")
    if (n, w, h) == (6_000_000, 3840, 2160):
        return base + " (BASELINE configs[4])"
    if (n, w, h) == (6_000_000, 1920, 1080):
        return base + " (stand-in for BASELINE configs[2]: no garden PLY ships with the reference or this container)"
    if (n, w, h) == (10_000, 256, 256):
        return base + " (BASELINE configs[0]'s size; scene kind S, not A)"
    return base + " (not a BASELINE config)"


def bin_count(w, h, bin_tiles):
    """Bins of bin_tiles x bin_tiles tiles on the screen (gs_frame_stats.bin_tiles: 8 by default, 4 when refined, larger
    only beyond 4K to keep the grid within 32 x 32)."""
    tx, ty = (w + 15) // 16, (h + 15) // 16
    return ((tx + bin_tiles - 1) // bin_tiles) * ((ty + bin_tiles - 1) // bin_tiles)


def algorithmic_bytes(n, v, d, e1, t, p, bins, bin_local=True):
    """HBM bytes per frame and pass that THIS decomposition has to move, whatever the kernels do
    internally.  n Gaussians, v visible, d tile instances, e1 (bin, Gaussian) candidates, t tiles, p pixels, bins of the
    level-1 grid.  bin_local: the depth order is taken inside k_bin_build; otherwise four global passes over V precede."""
    blocks = (n + 1023) // 1024                                   # level-1 blocks of 1024 items
    table = 4 * bins * blocks                                     # hist[bin][block]
    order = 0 if bin_local else 4 * (4 + 16) * v                  # 4 passes over V: histogram read 4 + (key, id) in 8 + out 8
    items = 4 * n + 8 * v if bin_local else (4 + 4 + 8) * v       # tiles + boxes of the items (+ the order on the global path)
    return {
        # pos 12 + cov3d 24 per Gaussian; opacity 4 + SH 192 per visible; 52 B of attributes out; tiles 4
        "preprocess": n * (12 + 24) + v * (4 + 192) + v * 52 + n * 4,
        # level-1 count + scan: the items in, the table out, then read and rewritten as prefixes
        "prefix_sum": items + 3 * table,
        # level-1 scatter: the items and the table in, one id per candidate out
        "preprocess_sort": items + table + 4 * e1,
        # (global depth order) + k_bin_build: id + depth + box per candidate in, one id per instance and the ranges out
        "sort": order + (4 + 4 + 8) * e1 + 4 * d + 8 * t,
        "tile_boundary": 0,                                       # produced inside k_bin_build
        "render": 40 * d + 16 * p,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=int(os.environ.get("GS_BENCH_N", 1_000_000)))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames-in-flight", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene", choices=["S", "T"], default="S", help="S: SURVEY 8d's synthetic scene; T: trained-scene statistics (synth.py)")
    ap.add_argument("--bgra8", action="store_true", help="also write the B8G8R8A8_UNORM image")
    ap.add_argument("--bgra8-only", action="store_true",
                    help="write ONLY the B8G8R8A8_UNORM image -- the reference's actual target (render.comp:98, Swapchain.cpp:22-28)")
    ap.add_argument("--exact", action="store_true",
                    help="the bit-identical blend (exp mode 2: libm's expf restated in binary64) as the benched mode")
    ap.add_argument("--fast-blend", action="store_true",
                    help="opt-in fast blend as the benched mode: polynomial exp + the FMA contractions GLSL permits")
    ap.add_argument("--hw-exp", action="store_true", help="opt-in: the hardware's v_exp_f32 + contractions as the benched mode")
    ap.add_argument("--sh16", action="store_true", help="opt-in binary16 SH storage (gs_scene_quantize_sh)")
    ap.add_argument("--ply", default=os.environ.get("GS_SCENE", ""),
                    help="bench this PLY (a trained scene, BASELINE configs[2]) instead of the synthetic one; also GS_SCENE=FILE")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the `other_configs` block (configs[4] and T(6e6) after the headline)")
    ap.add_argument("--dump-frames", default="", help="directory: every rank saves the frame of its pose (frame_rank<r>.npy) after the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_rank %= torch.cuda.device_count()  # ranks that are shown a subset of the node's GPUs (HIP_VISIBLE_DEVICES)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("GS_BENCH_BACKEND", "nccl")  # nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU
        dist.init_process_group(backend, device_id=dev if backend == "nccl" else None)

    pkg = entry.load_package()
    n, w, h = args.gaussians, args.width, args.height

    # ---- scene: built (or loaded: --ply / GS_SCENE) on rank 0, broadcast as one packed SoA blob (59 floats / Gaussian) ----
    if args.ply:  # a trained scene given at run time (BASELINE configs[2]): its size comes from the file
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        scene0 = None
        if rank == 0:
            scene0 = pkg.Scene.load_ply(args.ply, device=local_rank)  # GSScene::load (GSScene.cpp:26-68)
            cnt[0] = scene0.num_vertices
        if world > 1:
            dist.broadcast(cnt, src=0)
        n = int(cnt.item())
    blob = torch.empty(pkg.dist.blob_floats(n), dtype=torch.float32, device=dev)
    if rank == 0:
        if not args.ply:
            rec = pkg.synth.synth_records(n, seed=0, kind=args.scene)
            scene0 = pkg.Scene.from_records(rec, device=local_rank)  # GSScene::load path (activations on host)
            del rec
        src, floats = scene0.blob()
        assert floats == blob.numel()
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(blob.data_ptr()), ctypes.c_void_p(src), ctypes.c_size_t(floats * 4),
                           ctypes.c_int(3))  # hipMemcpyDeviceToDevice
        assert rc == 0
        scene0.close()
    torch.cuda.synchronize()
    t_bc = time.perf_counter()
    pkg.dist.broadcast_blob(blob, src=0)  # RCCL over xGMI; no-op at world == 1
    torch.cuda.synchronize()
    broadcast_ms = 1e3 * (time.perf_counter() - t_bc)
    rccl = rccl_evidence(torch, dist, dev, blob, rank, world, broadcast_ms) if world > 1 else None
    scene = pkg.Scene.from_device_blob(blob.data_ptr(), n, device=local_rank, keepalive=blob)
    if args.sh16:
        scene.quantize_sh()
    rend = pkg.Renderer(scene)
    rend.set_frames_in_flight(args.frames_in_flight)
    mode = "hw_exp" if args.hw_exp else ("fast" if args.fast_blend else ("exact" if args.exact else "default"))

    def set_mode(m):
        rend.set_exp_mode(MODES[m][0])
        rend.set_blend_contraction(MODES[m][1])
    set_mode(mode)

    cam = pkg.make_camera(rotation=pkg.dist.pose_quaternion(rank))  # pose k = default camera yawed k*5 deg
    u = pkg.camera_uniforms(cam, w, h)
    # one output image per frame in flight (frames on different streams must not share a target)
    fif = args.frames_in_flight
    want8 = args.bgra8 or args.bgra8_only
    outs = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) if not args.bgra8_only else None for _ in range(fif)]
    outs8 = [torch.empty((h, w, 4), dtype=torch.uint8, device=dev) if want8 else None for _ in range(fif)]

    def submit(i):
        o, o8 = outs[i % fif], outs8[i % fif]
        rend.render(u, o.data_ptr() if o is not None else 0, o8.data_ptr() if o8 is not None else 0)

    def sync_all():
        rend.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        submit(i)
    sync_all()
    # the W warm-up steps above are the contract's; the clocks of an idle chip need longer than a few milliseconds to
    # come up (round 2: the first timed batch ran at a ninth of the median) and the renderer's blend tuner measures its two
    # schedules over its first ~130 frames, so untimed K-step batches follow until (a) three consecutive ones lie within 3 % of
    # the RUNNING MINIMUM without improving on it -- round 5 compared neighbours, which a smooth ramp satisfies: the driver's first
    # five timed batches were 5-19 % slow -- (b) the tuner has settled, (c) MIN_WARM_SECONDS have passed (every rank runs the same
    # count: decided on the max over ranks)
    # The harness must not stall the frame loop it times: rounds 2-4 lost one 35-45 ms batch per run to Python's full garbage
    # collection (cyclic garbage from numpy's `ctypes.data_as` in the binding, 170 000 objects to walk once torch is imported;
    # profiles/r05_stall_hunt.txt).  The binding no longer produces garbage; what setup left behind is collected now and the
    # survivors are moved out of the collector's sight, so nothing count-triggered can land inside the timed region.  This happens
    # BEFORE the warm-up batches, not between them and the timed ones (round 6: the collection itself idles the GPU for tens of
    # milliseconds, the clocks fall, and the first timed batches ran 5-20 % slow however long the warm-up had been).
    gc.collect()
    gc.freeze()
    best, stable, warm_batches, tuner_settled, warm_s = None, 0, 0, False, 0.0
    for _ in range(400):
        sync_all()
        tw = time.perf_counter()
        for i in range(args.steps):
            submit(i)
        rend.synchronize()
        dtw = time.perf_counter() - tw
        settled = 1.0 if rend.blend_lockstep()[1] else 0.0
        if world > 1:
            t = torch.tensor([dtw, -settled], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtw, settled = float(t[0].item()), -float(t[1].item())
        warm_batches += 1
        warm_s += dtw
        # stable = this batch neither beats the minimum of the batches BEFORE it by more than 1 % (a ramp still going down: every
        # batch of a falling ramp is its own running minimum, which the first form of this test let pass) nor lies 3 % above it
        stable = stable + 1 if best is not None and 0.99 * best <= dtw <= 1.03 * best else 0
        best = dtw if best is None else min(best, dtw)
        tuner_settled = settled > 0
        if stable >= 4 and tuner_settled and warm_s >= MIN_WARM_SECONDS:
            break
    sync_all()
    rend.timing_totals(reset=True)
    rend.frame_intervals(reset=True)
    def timed_batch():
        """EXACTLY K frames between barrier + synchronize on both sides; max over ranks."""
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            submit(i)
        rend.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        own_s.append(dt)  # this rank's own wall time of the batch
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            dist.barrier()
        return dt

    # K frames of this workload take a few milliseconds: one such region is at the mercy of a single scheduling
    # hiccup.  The K-frame batch is therefore repeated until >= MIN_TIMED_SECONDS have been timed (every rank takes
    # the same count: it is derived from the max-over-ranks time of the first batch) and the MEDIAN batch is reported.
    own_s = []
    batch_s = [timed_batch()]
    repeats = max(1, min(200, int(math.ceil(MIN_TIMED_SECONDS / max(batch_s[0], 1e-6)))))
    for _ in range(repeats - 1):
        batch_s.append(timed_batch())
    elapsed = float(np.median(batch_s))

    sums, frames = rend.timing_totals(reset=True)
    intervals = rend.frame_intervals(reset=True)  # completion-to-completion, GPU timestamps, this rank's K frames
    st = rend.stats()
    lockstep = rend.blend_lockstep()  # what the renderer's own measurement chose for this workload (gs_set_blend_lockstep)

    # diagnostic only: the same frame with ONE frame in flight, so that per-pass spans are not stretched by
    # the other streams' kernels (the timed region above overlaps frames; its spans include that contention)
    rend.set_frames_in_flight(1)
    ts = time.perf_counter()
    for i in range(100):
        submit(i)
    rend.synchronize()
    serial_fps = 100 / (time.perf_counter() - ts)
    ssum, sframes = rend.timing_totals(reset=True)
    # diagnostic only: the same region with the blend in the other modes (this rank) -- what the opt-in relaxations are
    # worth on this box, and what the reference-exact default costs
    def mode_fps(m):
        rend.set_frames_in_flight(args.frames_in_flight)
        set_mode(m)
        for i in range(2 * args.steps):
            submit(i)
        rend.synchronize()
        best = []
        for _ in range(5):
            ta = time.perf_counter()
            for i in range(args.steps):
                submit(i)
            rend.synchronize()
            best.append(args.steps / (time.perf_counter() - ta))
        return float(np.median(best))
    alt = {m: (mode_fps(m) if m != mode else None) for m in MODES}
    # the benched frame in every mode, for the parity block (compared with the reference text in the cpu_baseline leg, where
    # that image exists anyway), and what the guard of the default mode did on it
    frames_for_parity = {}
    set_mode("default")
    rend.set_frames_in_flight(1)
    rend.render_host(u)
    gst = rend.stats()
    guard = {"break_decisions_resolved_exactly": int(gst.blend_resolved), "quadrants_rerendered_exactly": int(gst.blend_redo),
             "quadrants": ((w + 7) // 8) * ((h + 7) // 8),
             "what": "render.comp:83 decisions that fell inside the guard's proven window around 1e-4 and were taken from an exact "
                     "per-pixel replay; 8x8-pixel quadrants the fast pass abandoned and re-rendered with exp mode 2's arithmetic"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        for m in MODES:
            set_mode(m)
            frames_for_parity[m] = rend.render_host(u)[0]
    set_mode(mode)
    rend.set_frames_in_flight(args.frames_in_flight)
    if args.dump_frames:  # tests: every rank's pose against the checker (tests/test_gpu_dist.py)
        os.makedirs(args.dump_frames, exist_ok=True)
        np.save(os.path.join(args.dump_frames, f"frame_rank{rank}.npy"), rend.render_host(u)[0])
    # every rank's own rate over the timed batches (its frames / its own wall time, before the max over ranks)
    rank_fps = None
    if world > 1:
        mine = torch.tensor([args.steps / float(np.median(own_s))], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_fps = [round(float(t.item()), 2) for t in allr]
    if rank == 0:
        head = headline_numbers(world, args.steps, elapsed)
        T = ((w + 15) // 16) * ((h + 15) // 16)
        bin_edge = int(st.bin_tiles)
        bins = bin_count(w, h, bin_edge)
        nbytes = algorithmic_bytes(st.num_gaussians, st.num_visible, st.num_instances, st.num_bin_entries, T, w * h,
                                   bins, bin_local=int(st.sort_path) == 2)
        names = ["preprocess", "prefix_sum", "preprocess_sort", "sort", "tile_boundary", "render"]
        ms = {k: getattr(sums, "ms_" + k) / max(frames, 1) for k in names}
        per_pass = {k: {"ms": round(ms[k], 4), "alg_MB": round(nbytes[k] / 1e6, 2),
                        "GBps": round(nbytes[k] / 1e9 / (ms[k] * 1e-3), 1) if ms[k] > 0 else None} for k in names}
        # dominant kernel = largest share of a frame's GPU time when frames run one at a time (k_blend here);
        # its duration for the roofline is the span measured inside the timed (overlapped) region
        serial = {k: getattr(ssum, "ms_" + k) / max(sframes, 1) for k in names}
        dom = max(names, key=lambda k: serial[k])
        # run-to-run spread of the batches: interquartile range over the median (a single slow batch -- typically the
        # first one, while the clocks come up -- shows in batch_ms.max, not here)
        spread = (float(np.percentile(batch_s, 75) - np.percentile(batch_s, 25)) / elapsed) if len(batch_s) > 1 else None
        result = {
            # BASELINE.json's metric (its first clause; per-pass ms and HBM GB/s are `passes` and `roofline`); other
            # workloads (--gaussians / --width / --height) are named for what they are
            "metric": f"frames/sec at {w}\u00d7{h}, {n / 1e6:g}M Gaussians",
            "value": head["value"],
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            # every timed frame over every timed second (value is the MEDIAN batch): what a consumer that runs for minutes gets
            "sustained_frames_per_s": round(world * args.steps * len(batch_s) / float(np.sum(batch_s)), 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_name(n, w, h, world, args.scene, args.ply), "scene": "ply" if args.ply else args.scene,
                       "blend": BLEND_TEXT[mode], "blend_guard": guard,
                       "blend_lockstep": {"on": bool(lockstep[0]), "settled": bool(lockstep[1]),
                                          "what": "a tile's four waves take every chunk of its list together (their record gathers meet in L1); "
                                                  "automatic: the renderer measures both settings and keeps the faster -- frames are bit-identical either way"},
                       "gaussians": int(st.num_gaussians), "visible": int(st.num_visible),
                       "instances": int(st.num_instances), "bin_entries": int(st.num_bin_entries), "tiles": T, "output": "bgra8" if args.bgra8_only else ("rgba32f" + ("+bgra8" if args.bgra8 else "")),
                       "frames_in_flight": args.frames_in_flight, "parallelism": f"pose-sharded x{world}",
                       "depth_order_path": {1: "global", 2: "bin-local"}.get(int(st.sort_path), "?"),
                       "max_bin_entries": int(st.max_bin_entries), "bins": bins, "bin_tiles": bin_edge,
                       "sort_level": int(st.sort_level)},
            # the K-step region is timed `batches` times (each bracketed by barrier + synchronize); value / ms_per_step
            # are the median batch, spread = interquartile range / median over the batches
            "timed": {"batches": len(batch_s), "seconds": round(float(np.sum(batch_s)), 6),
                      # what the untimed warm-up did before the first timed batch: K-step batches until three in a row lay within
                      # 3 % of the running minimum AND the blend tuner had made its choice (a ramp cannot pass; verdict r5 item 7)
                      "warmup_batches": warm_batches, "tuner_settled_before_first_batch": bool(tuner_settled),
                      "first_batch_over_median": round(batch_s[0] / elapsed, 4),
                      "batch_ms": {"min": round(1e3 * min(batch_s), 4), "median": round(1e3 * elapsed, 4),
                                   "max": round(1e3 * max(batch_s), 4), "each": [round(1e3 * b, 2) for b in batch_s]},
                      "spread": round(spread, 4) if spread is not None else None,
                      # batches beyond 1.5 x the median (index, ms): mid-run hiccups the median hides and the line records
                      "outliers": [[i, round(1e3 * b, 2)] for i, b in enumerate(batch_s) if b > 1.5 * elapsed]},
            # intervals between the COMPLETIONS of consecutive frames (GPU timestamps), rank 0.  With frames in flight on several
            # streams completions come in bursts and out of order: this is what a consumer polling for finished frames sees, NOT a
            # per-frame time -- ms_per_step is that
            "completion_interval_ms": ({"p5": round(float(np.percentile(intervals, 5)), 4), "p50": round(float(np.percentile(intervals, 50)), 4),
                                        "p95": round(float(np.percentile(intervals, 95)), 4), "n": int(len(intervals))}
                                       if len(intervals) else None),
            "gpu_ms_per_frame": round(sums.ms_total / max(frames, 1), 4),
            "frames_per_s_one_in_flight": round(serial_fps, 2),  # diagnostic: one frame at a time (latency-bound)
            # diagnostics, this rank: the same region in the other blend modes (None = the benched one)
            "frames_per_s_default": round(alt["default"], 2) if alt["default"] else None,
            "frames_per_s_exact": round(alt["exact"], 2) if alt["exact"] else None,   # the bit-identical mode (round 3's default)
            "frames_per_s_fast_blend": round(alt["fast"], 2) if alt["fast"] else None,
            "frames_per_s_hw_exp": round(alt["hw_exp"], 2) if alt["hw_exp"] else None,
            "passes": per_pass,
            "passes_serial_ms": {k: round(getattr(ssum, "ms_" + k) / max(sframes, 1), 4) for k in names + ["total"]},
            "roofline": roofline(pkg, dom, workload_key(n, w, h, "ply" if args.ply else args.scene), MODES[mode], nbytes[dom], ms[dom],
                                 serial[dom], 1e3 * elapsed / args.steps),
        }
        if world > 1:
            result["rccl"] = dict(rccl or {}, per_rank_frames_per_s=rank_fps,
                                  slowest_rank=int(np.argmin(rank_fps)) if rank_fps else None)
        if args.bgra8_only:
            result["passes"]["render"]["alg_MB"] = round((nbytes["render"] - 12 * w * h) / 1e6, 2)  # 4 B per pixel out, not 16
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N = 1 only
            result["cpu_baseline"], result["parity"] = cpu_baseline(n, w, h, args.scene, frames_for_parity, ply=args.ply)
        # the other BASELINE configs a single command can show (verdict r3 item 8): after the headline, its renderer released
        if world == 1 and not args.no_other_configs and not args.ply and (n, w, h, args.scene) == (1_000_000, 1920, 1080, "S"):
            rend.close()
            scene.close()
            del blob, outs, outs8
            torch.cuda.empty_cache()
            result["other_configs"] = other_configs(pkg, torch, dev, local_rank, args, float(os.environ.get("GS_OTHER_CONFIGS_SECONDS", 75)))
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)


def blend_kernel_name(mode):
    """The k_blend<EXP, CONTRACT, GUARD> instantiation a (exp mode, contraction) pair launches, as rocprofv3 prints it."""
    exp, contract = mode
    if exp == 3:  # the guarded v_exp_f32 (with the contractions on there is nothing to guard: mode 1)
        return "k_blend<1, false, true>" if not contract else "k_blend<1, true, false>"
    return f"k_blend<{exp}, {'true' if contract else 'false'}, false>"


def rccl_evidence(torch, dist, dev, blob, rank, world, broadcast_ms):
    """What the collective actually saw (verdict r3 item 7: nobody has watched RCCL run with N > 1): an all-reduce of ones
    over the group (= the number of ranks that took part), the backend and its version, the broadcast's wall time and size,
    and a checksum of every rank's copy of the blob against rank 0's."""
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(ones)
    # checksum of the received blob: sum of its bit patterns (int64, wraps identically everywhere)
    chk = blob.view(torch.int32).to(torch.int64).sum().reshape(1)
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    sums = [int(t.item()) for t in allc]
    tmax = torch.tensor([broadcast_ms], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    backend = dist.get_backend()
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001 -- evidence only
            version = f"unavailable: {e}"
    mb = blob.numel() * 4 / 1e6
    return {"ranks": int(round(float(ones.item()))), "world_size": world, "backend": backend + (" (RCCL)" if backend == "nccl" else ""),
            "version": version, "blob_MB": round(mb, 1), "broadcast_ms": round(float(tmax.item()), 3),
            "broadcast_GBps": round(mb / 1e3 / (float(tmax.item()) * 1e-3), 1) if tmax.item() > 0 else None,
            "blob_checksums_equal_rank0": all(c == sums[0] for c in sums), "blob_checksum": sums[0]}


def other_configs(pkg, torch, dev, device, args, budget_s):
    """BASELINE configs[4] -- S(6e6) at 3840x2160 -- and the trained-like stand-in for configs[2] -- T(6e6) at 1920x1080 -- at
    their full sizes, in the same command as the headline: frames/s in the default mode (3 frames in flight, RGBA32F in HBM),
    serial per-pass ms, the exact mode's rate, and parity: the exact mode's frame bit for bit and the default mode's max abs
    against the CPU checker (oracle port, which tests pin to the reference text) on the same scene.  Time-boxed: a config
    that does not fit in what is left of `budget_s` is reported as skipped."""
    import concurrent.futures as cf
    oracle = entry.load_oracle()
    t_start = time.perf_counter()
    out = {}
    for name, (n, w, h, kind, label) in {"E": (6_000_000, 3840, 2160, "S", "BASELINE configs[4]"),
                                          "T": (6_000_000, 1920, 1080, "T", "trained-scene statistics; stand-in for BASELINE configs[2]")}.items():
        left = budget_s - (time.perf_counter() - t_start)
        if left < 25:
            out[name] = {"skipped": f"{left:.0f} s left of the {budget_s:.0f} s box"}
            continue
        t0 = time.perf_counter()
        # the scene, generated in slices on a thread pool (numpy releases the GIL in its array kernels; bit-identical to one call)
        chunk = 250_000
        with cf.ThreadPoolExecutor(min(24, os.cpu_count() or 8)) as ex:
            parts = list(ex.map(lambda s0: pkg.synth.synth_records(min(chunk, n - s0), seed=0, kind=kind, n_total=n, start=s0),
                                range(0, n, chunk)))
        rec = np.concatenate(parts)
        del parts
        t_gen = time.perf_counter() - t0
        scene = pkg.Scene.from_records(rec, device=device)
        rend = pkg.Renderer(scene)
        u = pkg.camera_uniforms(pkg.make_camera(), w, h)
        fif = args.frames_in_flight
        outs = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) for _ in range(fif)]

        def rate(mode, frames):
            rend.set_exp_mode(MODES[mode][0])
            rend.set_blend_contraction(MODES[mode][1])
            rend.set_frames_in_flight(fif)
            for i in range(max(200, frames // 2)):  # warm: depth-order level settled, the blend's lockstep measured and chosen (<= ~180 frames)
                rend.render(u, outs[i % fif].data_ptr(), 0)
            rend.synchronize()
            best = []
            for _ in range(3):
                ta = time.perf_counter()
                for i in range(frames):
                    rend.render(u, outs[i % fif].data_ptr(), 0)
                rend.synchronize()
                best.append(frames / (time.perf_counter() - ta))
            return float(np.median(best))
        fps = rate("default", 120)
        fps_exact = rate("exact", 120)
        rend.set_exp_mode(MODES["default"][0])
        lockstep = rend.blend_lockstep()
        rend.set_frames_in_flight(1)
        rend.timing_totals(reset=True)
        ts = time.perf_counter()
        for i in range(40):
            rend.render(u, outs[0].data_ptr(), 0)
        rend.synchronize()
        serial_fps = 40 / (time.perf_counter() - ts)
        ssum, sframes = rend.timing_totals(reset=True)
        img_default = rend.render_host(u)[0]
        st = rend.stats()
        rend.set_exp_mode(MODES["exact"][0])
        img_exact = rend.render_host(u)[0]
        entry_ = {"workload": workload_name(n, w, h, 1, kind), "label": label, "frames_per_s": round(fps, 2), "frames_per_s_exact": round(fps_exact, 2),
                  "gaussians": int(st.num_gaussians), "visible": int(st.num_visible), "instances": int(st.num_instances),
                  "sort_level": int(st.sort_level), "bin_tiles": int(st.bin_tiles), "frames_in_flight": fif,
                  "frames_per_s_one_in_flight": round(serial_fps, 2),
                  "blend_lockstep": {"on": bool(lockstep[0]), "settled": bool(lockstep[1]), "what": "gs_set_blend_lockstep(-1): measured by the renderer"},
                  "passes_serial_ms": {k: round(getattr(ssum, "ms_" + k) / max(sframes, 1), 4)
                                       for k in ("preprocess", "prefix_sum", "preprocess_sort", "sort", "tile_boundary", "render", "total")},
                  "blend_guard": {"break_decisions_resolved_exactly": int(st.blend_resolved), "quadrants_rerendered_exactly": int(st.blend_redo)},
                  "scene_generation_s": round(t_gen, 1)}
        rend.close()
        scene.close()
        del outs
        torch.cuda.empty_cache()
        if budget_s - (time.perf_counter() - t_start) > 12:  # parity against the CPU checker on this very scene
            verts = oracle.activate_records(rec)
            del rec
            ref_img, _ = oracle.render_frame(verts, oracle.cov3d(verts), oracle.camera_uniforms(oracle.default_camera(), w, h), want_image=True)
            d = np.abs(img_default[..., :3].astype(np.float64) - ref_img[..., :3])
            entry_["parity"] = {"against": "oracle port, default (reference) reading -- pinned bit for bit to the reference text by tests/test_oracle_vs_ref.py",
                                "exact_mode_bit_identical": bool(np.array_equal(img_exact.view(np.uint32), ref_img.view(np.uint32))),
                                "default_mode_max_abs": float(d.max()), "default_mode_pixels_above_1e-5": int((d.max(axis=2) > 1e-5).sum())}
            del verts, ref_img
        else:
            del rec
            entry_["parity"] = {"skipped": "time box"}
        entry_["seconds"] = round(time.perf_counter() - t0, 1)
        out[name] = entry_
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


def committed_counters(pkg, pass_name, wkey, mode):
    """Per-launch PMC counters of the dominant kernel from the newest committed rocprofv3 counter run
    (profiles/rNN_pmc_hbm_traffic.json: per workload, FETCH_SIZE, WRITE_SIZE and the SQ counters collected in separate
    passes by tools/profile_lite.sh).  Counters cannot be collected from inside this process, so the file is only trusted
    when it was collected from THIS library: it carries the hash of the kernel sources it profiled, and anything
    else -- a workload it does not hold, a kernel edited since -- yields (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files:
        return None, "no committed counter run"
    try:
        with open(files[-1]) as f:
            prof = json.load(f)
        name = os.path.basename(files[-1])
        if prof.get("library_source_sha256") != pkg.binding.library_source_hash():
            return None, f"{name} was collected from other kernel sources than the ones this library is built from"
        wl = prof.get("workloads", {}).get(wkey)
        if wl is None:
            return None, f"{name} holds no counters for workload {wkey}"
        want = {"render": blend_kernel_name(mode), "preprocess": "k_preprocess"}[pass_name]
        k = wl["kernels"][want]
        return {"file": "profiles/" + name, "kernel": want,
                "traffic": int((k["fetch_kb"] * k.get("fetch_scale", 1.0) + k["write_kb"]) * 1024),
                "valu_wave_insts": int(k["valu_wave_insts"])}, None
    except (OSError, KeyError, ValueError) as e:
        return None, f"unreadable counter file: {e}"


def committed_blend_work(wkey):
    """(pixel, entry) pairs the reference's loop walks on this workload (profiles/rNN_blend_work.json, counted once with
    the instrumented build: tools/blend_stats.py).  A property of the workload -- scene, camera, resolution -- not of
    the kernels, so no source hash is involved."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_blend_work.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                wl = json.load(f).get(wkey)
            if wl:
                return dict(wl, file="profiles/" + os.path.basename(path))
        except (OSError, ValueError):
            pass
    return None


def roofline(pkg, pass_name, wkey, mode, alg_bytes, ms_timed, ms_serial, ms_per_frame):
    """Roofline of the dominant kernel, on the wall time of ONE FRAME in the timed region (ms_per_step): with frames in flight
    the HIP-event span of a launch overlaps the other frames' kernels and spans are not additive, so work / frame time is
    the rate the chip sustains for this kernel alongside everything else a frame needs.  `one_in_flight` = the same launch
    with the GPU to itself.

    For k_blend (FP32 VALU-bound) the headline `frac` is SURVEY 8d's FLOP VIEW: 22 flop per (pixel,
    entry) pair the REFERENCE's loop walks on this workload (profiles/rNN_blend_work.json: a property of the scene, camera and
    resolution, not of these kernels) over the frame time, against the 157.3 TFLOP/s FP32 vector peak -- a figure a slower
    kernel cannot raise.  Beside it: `valu_issue`, the occupancy of the VALU issue slots (SQ_INSTS_VALU of this very library's
    kernel, from the committed counter run, over the frame time against 1228.8 G wave64-instructions/s) -- its numerator is
    the kernel's OWN instruction count, so it says how busy the pipe is, not how much useful work gets done -- and `hbm`, the
    algorithmic bytes against 8 TB/s.  Other kernels (k_preprocess at the 6 M configs) get the HBM view as the headline."""
    kernel = blend_kernel_name(mode) if pass_name == "render" else pass_name

    def rate(x, ms):
        return x / (ms * 1e-3) if ms and ms > 0 else None
    gb = {k: rate(alg_bytes / 1e9, ms) for k, ms in (("frame", ms_per_frame), ("span", ms_timed), ("serial", ms_serial))}
    hbm = {"achieved": round(gb["frame"], 1) if gb["frame"] else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gb["frame"] / HBM_PEAK_GBS, 4) if gb["frame"] else None, "algorithmic_bytes": int(alg_bytes),
           "one_in_flight": {"ms": round(ms_serial, 4), "achieved": round(gb["serial"], 1) if gb["serial"] else None,
                             "frac": round(gb["serial"] / HBM_PEAK_GBS, 4) if gb["serial"] else None}}
    c, why = committed_counters(pkg, pass_name, wkey, mode)
    common = {"kernel": c["kernel"] if c else kernel,
              "basis": "per_frame_time: one launch over ms_per_step (frames in flight overlap: spans are not additive)",
              "ms_per_frame": sig(ms_per_frame), "span_ms_in_timed_region": round(ms_timed, 4),
              "algorithmic_bytes": int(alg_bytes), "traffic": c["traffic"] if c else None,
              "counters": c["file"] if c else None}
    valu_issue = None
    if c is not None and pass_name == "render" and ms_per_frame > 0:
        insts = c["valu_wave_insts"]
        r, r1 = rate(insts, ms_per_frame), rate(insts, ms_serial)
        valu_issue = {"achieved": round(r / 1e9, 2), "peak": round(VALU_PEAK / 1e9, 2), "unit": "G wave64-inst/s", "frac": round(r / VALU_PEAK, 4),
                      "wave_insts": insts, "one_in_flight_frac": round(r1 / VALU_PEAK, 4) if r1 else None,
                      "sustained": {"peak": round(VALU_SUSTAINED / 1e9, 2), "frac": round(r / VALU_SUSTAINED, 4),
                                    "what": "issue rate tools/ubench/valu_rate.hip measures on this chip under dense VALU load (~1.85 GHz)"},
                      "note": "occupancy of the VALU issue slots: the numerator is this kernel's own SQ_INSTS_VALU, so a kernel that spends "
                              "more instructions on the same frame scores HIGHER here -- read `frac` (the flop view) for work done"}
    work = committed_blend_work(wkey) if pass_name == "render" else None
    if work and ms_per_frame > 0:
        fl = 22.0 * work["walked_pairs"]
        tf, tf1 = rate(fl / 1e12, ms_per_frame), rate(fl / 1e12, ms_serial)
        return dict(common, bound="valu", achieved=round(tf, 2), peak=FP32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / FP32_PEAK_TFLOPS, 4),
                    what="SURVEY 8d flop view: 22 flop x the (pixel, entry) pairs the reference's loop walks on this workload, per frame time, "
                         "against the FP32 vector peak (MFMA is not used: there is no dense contraction on this path)",
                    walked_pairs=work["walked_pairs"], contributing_pairs=work["contributing_pairs"], flop_per_pair=22, counts=work["file"],
                    one_in_flight={"ms": round(ms_serial, 4), "achieved": round(tf1, 2) if tf1 else None,
                                   "frac": round(tf1 / FP32_PEAK_TFLOPS, 4) if tf1 else None},
                    valu_issue=valu_issue, valu_issue_note=None if valu_issue else why, hbm=hbm)
    return dict(common, bound="hbm", achieved=hbm["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm["frac"],
                one_in_flight=hbm["one_in_flight"], valu_issue=valu_issue,
                note=("HBM view" + (f": {why}" if why else "") + ("; for k_blend the binding roof is FP32 VALU issue (DESIGN.md section 4): the "
                      "flop view needs the workload's walked-pair count (tools/blend_stats.py -> profiles/rNN_blend_work.json)"
                      if pass_name == "render" else "")))


def cpu_baseline(n, w, h, kind="S", gpu_frames=None, ply=None):
    """The oracle (CPU restatement of the reference shaders, all host cores via OpenMP) on one frame of
    the same workload -- a bounded sample (about 10-30 s of CPU work).  Baseline only.  Returns (cpu_baseline, parity):
    parity = the GPU's frames of this workload (gpu_frames: mode -> image) against the reference text's frame (or the
    port's, which tests pin to it bit for bit, when oracle/_ref did not travel)."""
    oracle = entry.load_oracle()
    pkg = entry.load_package()
    if ply:
        verts = oracle.load_ply(ply)  # the checker's own reader + activation (GSScene.cpp:26-68 restated)
    else:
        verts = oracle.activate_records(pkg.synth.synth_records(n, seed=0, kind=kind))
    cov = oracle.cov3d(verts)
    u = oracle.camera_uniforms(oracle.default_camera(), w, h)
    ref_img, _ = oracle.render_frame(verts, cov, u, want_image=True)  # warm-up (scalar blend = the parity checker)
    oracle.set_simd_blend(True)  # the baseline is timed with the AVX2 blend, which must reproduce the checker's image
    simd_img, _ = oracle.render_frame(verts, cov, u, want_image=True)
    assert np.array_equal(ref_img.view(np.uint32), simd_img.view(np.uint32)), "AVX2 baseline blend != scalar oracle"
    frames, t0, budget = 0, time.perf_counter(), float(os.environ.get("GS_CPU_BASELINE_SECONDS", 10))
    ms = np.zeros(6)
    while True:
        _, st = oracle.render_frame(verts, cov, u, want_image=True)
        frames += 1
        ms += np.array(st.ms)
        dt = time.perf_counter() - t0
        if dt >= budget or frames >= 64:
            break
    # the same frame on ONE core (SURVEY 8d / BASELINE.md section 3 ask for both): one frame is the bounded sample
    all_threads = oracle.num_threads()
    oracle.set_num_threads(1)
    t1 = time.perf_counter()
    _, st1 = oracle.render_frame(verts, cov, u, want_image=True)
    dt1 = time.perf_counter() - t1
    oracle.set_num_threads(all_threads)
    one_core = {"value": round(1.0 / dt1, 4), "unit": "frames/s", "cores": 1, "sample": f"1 frame in {dt1:.1f} s",
                "ms_per_pass": [round(x, 1) for x in st1.ms]}
    # the reference's OWN shader text compiled for the CPU (oracle/_ref; built where /root/reference is mounted and
    # shipped as a prebuilt file): one frame of the per-frame passes, scalar GLSL invocations, OpenMP over workgroups
    reference_text = None
    checker_img, checker = ref_img, "oracle port (default reading; pinned bit for bit to the reference text by tests/test_oracle_vs_ref.py)"
    try:
        ref = entry.load_ref()
        if ref.available():
            rcov = ref.cov3d(verts)
            tr = time.perf_counter()
            rst = ref.stages(verts, u, cov=rcov)
            dtr = time.perf_counter() - tr
            checker_img, checker = rst["image"], "reference text: src/shaders/*.comp compiled for the CPU (oracle/_ref), whole frame"
            reference_text = {"value": round(1.0 / dtr, 4), "unit": "frames/s", "cores": all_threads, "kind": "reference",
                              "sample": f"1 frame in {dtr:.1f} s: src/shaders/*.comp compiled for the CPU (oracle/build_ref.py), "
                                        "one scalar invocation per thread slot, prefix_sum.comp's ceil(log2 N)+1 passes as written, "
                                        + ("the 8 radix passes = sort/hist.comp + sort/sort.comp as written (workgroups emulated with one fiber per "
                                           "invocation: barriers, shared arrays, 32-wide subgroup operations)" if rst.get("text_sort")
                                           else "std::stable_sort in place of the 8 radix passes (instances beyond GS_REF_TEXT_SORT_MAX)"),
                              "max_abs_vs_port": float(np.abs(rst["image"] - ref_img).max())}
    except Exception as e:  # baseline garnish only: never fail the bench line over it
        reference_text = {"error": str(e)}
    parity = None
    if gpu_frames:
        parity = {"against": checker}
        for m, img in gpu_frames.items():
            d = np.abs(img[..., :3].astype(np.float64) - checker_img[..., :3]).max(axis=2)
            parity[m] = {"max_abs_vs_reference_text": float(d.max()), "pixels_above_1e-4": int((d > 1e-4).sum()),
                         "pixels_above_1e-5": int((d > 1e-5).sum()),
                         "bit_identical": bool(np.array_equal(img.view(np.uint32), checker_img.view(np.uint32)))}
    return {"value": round(frames / dt, 4), "unit": "frames/s", "cores": all_threads, "kind": "port", "one_core": one_core,
            "reference_text": reference_text,
            "sample": f"{frames} frame(s) of the same workload (N={n}, {w}x{h}, D={st.num_instances}) in {dt:.1f} s wall; "
                      "oracle = CPU restatement of the reference shaders, OpenMP over Gaussians/tiles, AVX2 blend (8 pixels per step), "
                      "sliced parallel LSD sort",
            "ms_per_pass": [round(x, 2) for x in (ms / frames)]}, parity


if __name__ == "__main__":
    main()
