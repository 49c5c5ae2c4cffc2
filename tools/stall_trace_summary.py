#!/usr/bin/env python3
"""Summaries for the stall hunt (tools/r05_stall.sh):
  stall_trace_summary.py rocprof DIR   -- every HIP / HSA API call longer than 2 ms in rocprofv3's csv traces, with its neighbours
  stall_trace_summary.py amdlog FILE   -- the largest gaps between consecutive AMD_LOG_LEVEL lines, with the lines around them
"""
import csv
import glob
import os
import re
import sys


def rocprof(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*_api_trace.csv"), recursive=True))
    kfiles = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                try:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Domain", ""), r["Function"], r.get("Thread_Id", "")))
                except (KeyError, ValueError):
                    pass
    rows.sort()
    print(f"{len(rows)} API calls in {[os.path.basename(f) for f in files]}")
    if not rows:
        return
    t0 = rows[0][0]
    long_calls = [(i, r) for i, r in enumerate(rows) if r[1] - r[0] > 2_000_000]
    print(f"{len(long_calls)} calls longer than 2 ms (time since the first call, duration, domain, function, thread):")
    for i, r in long_calls[:200]:
        print(f"  +{(r[0] - t0) * 1e-9:9.4f} s  {(r[1] - r[0]) * 1e-6:9.3f} ms  {r[2]:12s} {r[3]}  tid {r[4]}")
    # for the calls between 15 and 120 ms (the stall's size): what ran inside them (nested calls on any thread)
    for i, r in long_calls:
        dur = (r[1] - r[0]) * 1e-6
        if not 15.0 <= dur <= 120.0:
            continue
        print(f"--- inside {r[3]} (+{(r[0] - t0) * 1e-9:.4f} s, {dur:.2f} ms): calls > 0.2 ms nested in it")
        for q in rows[i + 1:i + 4000]:
            if q[0] > r[1]:
                break
            if q[1] - q[0] > 200_000:
                print(f"      +{(q[0] - r[0]) * 1e-6:8.3f} ms  {(q[1] - q[0]) * 1e-6:8.3f} ms  {q[2]:12s} {q[3]}  tid {q[4]}")
    # kernel timeline: the largest gaps between consecutive kernel starts
    ks = []
    for f in kfiles:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                try:
                    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
                except (KeyError, ValueError):
                    pass
    ks.sort()
    busy_until = 0
    gaps = []
    for i, k in enumerate(ks):
        if i and k[0] > busy_until:
            gaps.append((k[0] - busy_until, i))
        busy_until = max(busy_until, k[1])
    gaps.sort(reverse=True)
    print(f"{len(ks)} kernels; largest GPU-idle gaps (no kernel running):")
    for g, i in gaps[:8]:
        print(f"  {g * 1e-6:9.3f} ms idle before kernel #{i} {ks[i][2]} at +{(ks[i][0] - t0) * 1e-9:.4f} s")


def amdlog(path):
    pat = re.compile(r"^:(\d):([^:]+):\s*(\d+)\s*:\s*(\d+) us:")
    lines = []
    with open(path, errors="replace") as fh:
        for line in fh:
            m = pat.match(line)
            if not m:
                continue
            lines.append((int(m.group(4)), line.rstrip()[:260]))
    print(f"{len(lines)} timestamped lines")
    gaps = sorted(((lines[i + 1][0] - lines[i][0], i) for i in range(len(lines) - 1)), reverse=True)[:4]
    for g, i in gaps:
        print(f"=== gap of {g / 1000.0:.3f} ms after line {i} (+{(lines[i][0] - lines[0][0]) * 1e-6:.4f} s)")
        for j in range(max(0, i - 25), min(len(lines), i + 26)):
            print(f"   {'>>' if j == i + 1 else '  '} {lines[j][1]}")


if __name__ == "__main__":
    {"rocprof": rocprof, "amdlog": amdlog}[sys.argv[1]](sys.argv[2])
