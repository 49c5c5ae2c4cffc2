"""Turn the rocprofv3 CSVs collected by tools/profile_lite.sh into the summaries committed under profiles/.

    python tools/profile_summary.py rNN        (reads gpurun_out/prof_rNN/<workload>/..., writes profiles/rNN_*)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
NAMES = {"B": "config B (S(1e6), 1920x1080)", "C": "config C stand-in (S(6e6), 1920x1080)",
         "T": "config C stand-in with trained-scene statistics (T(6e6), 1920x1080)", "E": "config E (S(6e6), 3840x2160)"}


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].replace("gs::", "")


def bench_line(path):
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1])
    except (OSError, IndexError, ValueError):
        return None


def kernel_stats(wl, sub, out_name, header):
    files = glob.glob(os.path.join(src, wl, sub, "*_kernel_stats.csv"))
    if not files:
        print("missing", wl, sub)
        return
    rows = list(csv.DictReader(open(files[0])))
    b = bench_line(os.path.join(src, wl, sub, "bench.json"))
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("# " + header + "\n")
        try:
            ls = open(os.path.join(src, wl, "lockstep.txt")).read().strip()
            f.write("# blend lockstep pinned to the renderer's own measured choice for this workload (an unprofiled run, three frames in flight): %s\n"
                    % {"True": "on", "False": "off"}.get(ls, ls))
        except OSError:
            pass
        if b:
            f.write("# %s under the profiler: %s frames/s, %d frame(s) in flight; HIP-event spans (us) %s; config %s\n"
                    % (b["driver"], b["value"], b["frames_in_flight"], json.dumps(b["spans_us"]), json.dumps(b["config"])))
        f.write("%-34s %9s %14s %11s %8s %10s %10s\n" % ("kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"))
        for r in rows:
            f.write("%-34s %9d %14.1f %11.2f %8.2f %10.2f %10.2f\n"
                    % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                       float(r["Percentage"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    print("wrote", out_name)


def counters(wl, sub):
    files = glob.glob(os.path.join(src, wl, sub, "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        print("missing", wl, sub)
        return agg
    for r in csv.DictReader(open(files[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def med(v):
    """Per-dispatch figure of a kernel: the MEDIAN over its dispatches (a renderer's first frame runs at the smallest
    sort level, overflows it and is re-run: that frame's blend walks near-empty lists)."""
    if not v:
        return float("nan")
    w = sorted(v)
    m = len(w) // 2
    return w[m] if len(w) % 2 else 0.5 * (w[m - 1] + w[m])


os.makedirs(dst, exist_ok=True)
CMD = "python tools/tune_sweep.py --no-prime --batches 1 <workload args>"
workloads = {}
pmc_txt = []
for wl in [d for d in ("B", "C", "T", "E") if os.path.isdir(os.path.join(src, d))]:
    kernel_stats(wl, "serial", f"{tag}_kernel_stats_{wl}_serial.txt",
                 f"rocprofv3 --kernel-trace --stats --output-format csv -- {CMD} --fif 1   (MI355X, {NAMES[wl]}, one frame at a "
                 "time: clean per-kernel durations; default blend = exp mode 3, the guarded v_exp_f32)")
    if wl == "B":
        kernel_stats(wl, "serial_exact", f"{tag}_kernel_stats_{wl}_serial_exact.txt",
                     f"rocprofv3 --kernel-trace --stats --output-format csv -- {CMD} --fif 1 --exp-mode 2   (MI355X, {NAMES[wl]}, one frame at a "
                     "time, the bit-identical blend: exp mode 2)")
    kernel_stats(wl, "default", f"{tag}_kernel_stats_{wl}_3inflight.txt",
                 f"rocprofv3 --kernel-trace --stats --output-format csv -- {CMD} --fif 3   (MI355X, {NAMES[wl]}, 3 frames in "
                 "flight like the default bench command; durations include contention from the other frames in flight)")
    pmc, fetch, write = counters(wl, "pmc"), counters(wl, "fetch"), counters(wl, "write")
    fast = counters(wl, "pmc_fast")
    for sub in ("pmc_exact", "pmc_hw"):  # the other blend modes (config B): SQ counters only
        for k, v in counters(wl, sub).items():
            if k.startswith("k_blend"):
                fast[k] = v
    if not pmc:
        continue
    b = bench_line(os.path.join(src, wl, "serial", "bench.json"))
    cfg = b["config"] if b else {}
    n, v, d, w, h = (cfg.get(k, 0) for k in ("gaussians", "visible", "instances", "width", "height"))
    kernels = {}
    pmc_txt.append(f"## {NAMES[wl]}: N={n} V={v} D={d}")
    pmc_txt.append("%-34s %10s %10s %10s %12s %12s %10s %10s"
                   % ("kernel", "VALU_inst", "SALU_inst", "LDS_inst", "WAVE_CYCLES", "BUSY_CYCLES", "FETCH_KB", "WRITE_KB"))
    for k in sorted(pmc, key=lambda k: -med(pmc[k]["SQ_INSTS_VALU"])):
        c = pmc[k]
        pmc_txt.append("%-34s %10.4g %10.4g %10.4g %12.4g %12.4g %10.0f %10.0f"
                       % (k, med(c["SQ_INSTS_VALU"]), med(c["SQ_INSTS_SALU"]), med(c["SQ_INSTS_LDS"]), med(c["SQ_WAVE_CYCLES"]),
                          med(c["SQ_BUSY_CYCLES"]), med(fetch[k]["FETCH_SIZE"]), med(write[k]["WRITE_SIZE"])))
        kernels[k] = {"fetch_kb": round(med(fetch[k]["FETCH_SIZE"])), "write_kb": round(med(write[k]["WRITE_SIZE"])),
                      # gfx950: FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md); applied to the
                      # coalesced plane reads of k_preprocess only -- uncalibrated for 16-byte gathers
                      "fetch_scale": 2.0 if k.startswith("k_preprocess") else 1.0,
                      "valu_wave_insts": round(med(pmc[k]["SQ_INSTS_VALU"])), "salu_wave_insts": round(med(pmc[k]["SQ_INSTS_SALU"]))}
    for k in fast:  # the blend's other modes: SQ counters only
        if k.startswith("k_blend") and k not in kernels:
            base = kernels.get("k_blend<1, false, true>", {})
            kernels[k] = {"fetch_kb": base.get("fetch_kb", 0), "write_kb": base.get("write_kb", 0), "fetch_scale": 1.0,
                          "valu_wave_insts": round(med(fast[k]["SQ_INSTS_VALU"])), "salu_wave_insts": round(med(fast[k]["SQ_INSTS_SALU"])),
                          "note": "FETCH/WRITE taken from the default blend's run (same lists, same records)"}
            pmc_txt.append("%-34s %10.4g %10.4g %10.4g %12.4g %12.4g   (another blend mode; separate run)"
                           % (k, med(fast[k]["SQ_INSTS_VALU"]), med(fast[k]["SQ_INSTS_SALU"]), med(fast[k]["SQ_INSTS_LDS"]),
                              med(fast[k]["SQ_WAVE_CYCLES"]), med(fast[k]["SQ_BUSY_CYCLES"])))
    for k in kernels:
        if k.startswith("k_blend") and d:
            kernels[k]["algorithmic_bytes"] = 40 * d + 16 * w * h
        if k.startswith("k_preprocess") and n:
            kernels[k]["algorithmic_bytes"] = n * 40 + v * 248
    kind = cfg.get("scene", "S")
    workloads[f"{kind}({n})@{w}x{h}"] = {"name": NAMES[wl], "gaussians": n, "visible": v, "instances": d, "width": w, "height": h,
                                        "kernels": kernels}

if workloads:
    with open(os.path.join(dst, tag + "_pmc_counters.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <counters> -- " + CMD + " --fif 1 --frames 3 --warm 200  (MI355X; lockstep pinned to the renderer's measured choice); three separate runs per workload:\n"
                "#   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY | FETCH_SIZE | WRITE_SIZE\n"
                "# medians over the dispatches of a run.  FETCH_SIZE / WRITE_SIZE in KB.  SQ_*_CYCLES are quad-cycles; SQ_BUSY_CYCLES is summed over 32 shader engines.\n")
        f.write("\n".join(pmc_txt) + "\n")
    with open(os.path.join(dst, tag + "_pmc_hbm_traffic.json"), "w") as f:
        json.dump({"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate runs per workload) "
                               "-- " + CMD + " --fif 1 --frames 3 --warm 2 on MI355X.  KB per launch (median over the launches of "
                               "the run), raw counter values; fetch_scale is the gfx950 correction (FETCH_SIZE reports half of a wide "
                               "coalesced streaming read: applied to k_preprocess; gathers are left raw).  Infinity-Cache hits are "
                               "counted as traffic.  valu_wave_insts = SQ_INSTS_VALU per launch.",
                   "library_source_sha256": open(os.path.join(src, "source_hash.txt")).read().strip(),
                   "workloads": workloads}, f, indent=1)
    print("wrote", tag + "_pmc_counters.txt,", tag + "_pmc_hbm_traffic.json", list(workloads))
