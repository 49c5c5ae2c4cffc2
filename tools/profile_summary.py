"""Turn the rocprofv3 CSVs collected by tools/profile_round.sh into the summaries committed under profiles/.

    python tools/profile_summary.py rNN        (reads gpurun_out/prof_rNN, writes profiles/rNN_*)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def bench_line(path):
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1])
    except (OSError, IndexError, ValueError):
        return None


def kernel_stats(sub, out_name, header):
    files = glob.glob(os.path.join(src, sub, "*_kernel_stats.csv"))
    if not files:
        print("missing", sub)
        return
    rows = list(csv.DictReader(open(files[0])))
    b = bench_line(os.path.join(src, sub, "bench.json"))
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("# " + header + "\n")
        if b and b.get("driver"):  # tools/profile_lite.sh: the profiled process is tools/tune_sweep.py
            f.write("# %s under the profiler: %s frames/s, %d frame(s) in flight; HIP-event spans (us) %s\n"
                    % (b["driver"], b["value"], b["frames_in_flight"], json.dumps(b["spans_us"])))
        elif b:
            f.write("# bench line under the profiler: value=%s frames/s ms_per_step=%s; k_blend HIP-event span in the "
                    "timed region %.4f ms (one frame at a time: %.4f ms)\n"
                    % (b["value"], b["ms_per_step"], b["passes"]["render"]["ms"], b["passes_serial_ms"]["render"]))
        f.write("%-34s %9s %14s %11s %8s %10s %10s\n" % ("kernel", "calls", "total_us", "avg_us", "pct", "min_us", "max_us"))
        for r in rows:
            f.write("%-34s %9d %14.1f %11.2f %8.2f %10.2f %10.2f\n"
                    % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                       float(r["Percentage"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    print("wrote", out_name)


def counters(sub):
    files = glob.glob(os.path.join(src, sub, "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        print("missing", sub)
        return agg
    for r in csv.DictReader(open(files[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    """Per-dispatch figure of a kernel: the MEDIAN over its dispatches.  (A renderer's first frame runs at the smallest
    sort level, overflows it and is re-run: that frame's blend walks near-empty lists, and in a short counter run it
    would pull a mean down by a fifth.)"""
    if not v:
        return float("nan")
    w = sorted(v)
    m = len(w) // 2
    return w[m] if len(w) % 2 else 0.5 * (w[m - 1] + w[m])


os.makedirs(dst, exist_ok=True)
lite = (bench_line(os.path.join(src, "serial", "bench.json")) or {}).get("driver") is not None
if lite:
    CMD = "python tools/tune_sweep.py --no-prime --batches 1 --frames 300"
    PMC_CMD = "python tools/tune_sweep.py --no-prime --batches 1 --fif 1 --frames 3 --warm 1"
    kernel_stats("default", tag + "_kernel_stats_default.txt",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- " + CMD + " --fif 3   (MI355X, config B, 3 frames in "
                 "flight like the default bench command -- the same C-ABI calls as bench.py's timed region, no torch; durations "
                 "include contention from the other frames in flight)")
    kernel_stats("serial", tag + "_kernel_stats_serial.txt",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- " + CMD + " --fif 1   (MI355X, config B, one frame at a "
                 "time: clean per-kernel durations)")
else:
    PMC_CMD = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --frames-in-flight 1"
    kernel_stats("default", tag + "_kernel_stats_default.txt",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
                 "   (MI355X, config B, 3 frames in flight = the default bench command; durations include contention from the"
                 " other frames in flight; the 100 one-in-flight diagnostic launches are pooled in)")
    kernel_stats("serial", tag + "_kernel_stats_serial.txt",
                 "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline"
                 " --frames-in-flight 1   (MI355X, config B, one frame at a time: clean per-kernel durations)")
kernel_stats("configE", tag + "_kernel_stats_configE_serial.txt",
             "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline"
             " --frames-in-flight 1 --gaussians 6000000 --width 3840 --height 2160   (MI355X, config E, one frame at a time)")

pmc, fetch, write = counters("pmc"), counters("fetch"), counters("write")
if pmc:
    with open(os.path.join(dst, tag + "_pmc_counters.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <counters> -- " + PMC_CMD + "  (MI355X, config B); three separate runs:\n"
                "#   SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU "
                "SQ_WAIT_ANY | FETCH_SIZE | WRITE_SIZE\n"
                "# medians over the dispatches of the run.  FETCH_SIZE / WRITE_SIZE in KB.  SQ_*_CYCLES are quad-cycles; SQ_BUSY_CYCLES is summed "
                "over 32 shader engines.\n")
        f.write("%-30s %10s %10s %10s %12s %12s %10s %10s\n"
                % ("kernel", "VALU_inst", "SALU_inst", "LDS_inst", "WAVE_CYCLES", "BUSY_CYCLES", "FETCH_KB", "WRITE_KB"))
        order = sorted(pmc, key=lambda k: -mean(pmc[k]["SQ_INSTS_VALU"]))
        for k in order:
            c = pmc[k]
            f.write("%-30s %10.3g %10.3g %10.3g %12.4g %12.4g %10.0f %10.0f\n"
                    % (k, mean(c["SQ_INSTS_VALU"]), mean(c["SQ_INSTS_SALU"]), mean(c["SQ_INSTS_LDS"]),
                       mean(c["SQ_WAVE_CYCLES"]), mean(c["SQ_BUSY_CYCLES"]), mean(fetch[k]["FETCH_SIZE"]),
                       mean(write[k]["WRITE_SIZE"])))
    print("wrote", tag + "_pmc_counters.txt")

    b = bench_line(os.path.join(src, "bench_default.json")) or bench_line(os.path.join(src, "sweep_default.json"))
    cfg = b["config"] if b else {}
    n, v, d = cfg.get("gaussians", 0), cfg.get("visible", 0), cfg.get("instances", 0)
    kernels = {}
    for k in pmc:
        e = {"fetch_kb": round(mean(fetch[k]["FETCH_SIZE"])), "write_kb": round(mean(write[k]["WRITE_SIZE"])),
             # gfx950: FETCH_SIZE reports half of a wide coalesced stream (MI355X_MICROARCH.md); applied to the
             # coalesced plane reads of k_preprocess only -- uncalibrated for 16-byte gathers
             "fetch_scale": 2.0 if k == "gs::k_preprocess" else 1.0,
             "valu_wave_insts": round(mean(pmc[k]["SQ_INSTS_VALU"]))}
        kernels[k.replace("gs::", "")] = e
    if "k_blend" in kernels and d:
        kernels["k_blend"]["algorithmic_bytes"] = 40 * d + 16 * 1920 * 1080
    if "k_preprocess" in kernels and n:
        kernels["k_preprocess"]["algorithmic_bytes"] = n * 40 + v * 248
    with open(os.path.join(dst, tag + "_pmc_hbm_traffic.json"), "w") as f:
        json.dump({"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate runs) "
                               "-- " + PMC_CMD + " on MI355X, "
                               "config B.  KB per launch (median over the launches of the run), raw counter values; "
                               "fetch_scale is the gfx950 correction (FETCH_SIZE reports half of a wide coalesced "
                               "streaming read: applied to k_preprocess; the blend's 16-byte gathers are left raw). "
                               "Infinity-Cache hits are counted as traffic.  valu_wave_insts = SQ_INSTS_VALU per launch.",
                   "library_source_sha256": open(os.path.join(src, "source_hash.txt")).read().strip()
                   if os.path.exists(os.path.join(src, "source_hash.txt")) else None,
                   "gaussians": n, "width": 1920, "height": 1080, "kernels": kernels}, f, indent=1)
    print("wrote", tag + "_pmc_hbm_traffic.json")

for name in ("bench_default", "bench_configE", "bench_configC_standin", "bench_default_hwexp", "bench_configE_sh16"):
    b = bench_line(os.path.join(src, name + ".json"))
    if b:
        with open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w") as f:
            json.dump(b, f)
            f.write("\n")
        print("wrote", "%s_%s.json" % (tag, name), b["value"], b["unit"])
