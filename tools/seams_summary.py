"""Seams of a frame rendered alone, from a rocprofv3 --kernel-trace CSV: for every frame (k_preprocess .. k_blend on one stream) the
kernels' durations and the gap between the end of one and the start of the next; medians over the trace's frames.
    python tools/seams_summary.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import re
import sys

import numpy as np

files = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("gs::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
frames, cur = [], []
for s, e, n in rows:
    if n.startswith("k_preprocess"):
        cur = [(s, e, n)]
    elif cur:
        cur.append((s, e, n))
        if n.startswith("k_blend"):
            frames.append(cur)
            cur = []
frames = frames[len(frames) // 3:]  # the settled part of the run
shape = [n for _, _, n in frames[-1]]
same = [f for f in frames if [n for _, _, n in f] == shape]
dur = np.array([[e - s for s, e, _ in f] for f in same]) / 1e3
gap = np.array([[f[k + 1][0] - f[k][1] for k in range(len(f) - 1)] for f in same]) / 1e3
span = np.array([f[-1][1] - f[0][0] for f in same]) / 1e3
period = np.diff(np.array([f[0][0] for f in same])) / 1e3
between = np.array([same[k + 1][0][0] - same[k][-1][1] for k in range(len(same) - 1)]) / 1e3
print(f"frames {len(same)} of {len(frames)} with the kernel sequence of the last one")
print(f"{'kernel':40s} {'median us':>10s}   gap to the next kernel (end -> start), median us")
for k, n in enumerate(shape):
    g = f"{np.median(gap[:, k]):8.2f}" if k < len(shape) - 1 else ""
    print(f"{n[:40]:40s} {np.median(dur[:, k]):10.2f}   {g}")
print(f"sum of kernels {np.median(dur.sum(axis=1)):.2f} us; sum of the gaps inside a frame {np.median(gap.sum(axis=1)):.2f} us; first start -> last end {np.median(span):.2f} us")
print(f"frame period (start to start) {np.median(period):.2f} us = {1e6 / np.median(period):.0f} frames/s; gap between frames (blend end -> next preprocess start) {np.median(between):.2f} us")
