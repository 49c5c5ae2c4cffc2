#!/usr/bin/env python3
"""What a restructured blend would have to do -- counted on the CPU, before any kernel is written (verdict r5 item 2).

For a sample of tiles of a workload the checker's per-tile lists (the reference's order) are walked pixel by pixel in float64
(decisions as render.comp:66-85 takes them; rounding noise does not move counts) and the work of each candidate schedule is
counted in the units the kernel pays in -- wave iterations of the pair loop (one iteration = one entry evaluated by a wave,
~25 VALU instructions) and chunk classifications:

  quad      the shipped schedule: one wave per 8x8 quadrant, exact ellipse-vs-rectangle cull per (entry, quadrant), every kept
            entry evaluated by all 64 lanes until the quadrant's last pixel has saturated
  blocks    per-16-lane entry streams: the wave's four 4x4 blocks each walk their own compacted list; an iteration serves one
            entry PER BLOCK, the wave iterates max over its blocks of the blocks' list lengths (per 64-entry chunk);
            `ideal` culls a block exactly (any pixel of the block passes render.comp:68,78), `aabb` with the bounding box of
            the alpha-cut ellipse (what a 16-byte spare quarter of the record could carry)
  wide      two pixels per lane: one wave per 16x8 half tile, the exact cull on that rectangle; an iteration costs ~1.9 of
            `quad`'s (one record read and one scalar tail per two evaluations)

    python tools/blend_variants_sim.py [--tiles 160] [--workload B]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

WORKLOADS = {"B": (1_000_000, 1920, 1080, "S"), "T1": (1_000_000, 1920, 1080, "T"), "A": (10_000, 256, 256, "A")}


def min_q_rect(c00, c01, c11, u, v, xa, xb, ya, yb):
    """min over the rectangle of 0.5 (c00 dx^2 + c11 dy^2) + c01 dx dy, d = centre - pixel (vectorised over entries)."""
    dx_lo, dx_hi, dy_lo, dy_hi = u - xb, u - xa, v - yb, v - ya
    in_x = (dx_lo <= 0) & (dx_hi >= 0)
    in_y = (dy_lo <= 0) & (dy_hi >= 0)
    h00, h11 = 0.5 * c00, 0.5 * c11
    a = np.where(dx_lo > 0, dx_lo, dx_hi)
    t = np.clip(-c01 / c11 * a, dy_lo, dy_hi)
    qv = h00 * a * a + t * (h11 * t + c01 * a)
    b = np.where(dy_lo > 0, dy_lo, dy_hi)
    s = np.clip(-c01 / c00 * b, dx_lo, dx_hi)
    qh = h11 * b * b + s * (h00 * s + c01 * b)
    return np.where(in_x, np.where(in_y, 0.0, qh), np.where(in_y, qv, np.minimum(qv, qh)))


def simulate_tile(co, uv, cut, x0, y0, w, h):
    """co [n,4] conic+opacity, uv [n,2], cut [n] (alpha cut on power, <= 0), the tile's entries in list order."""
    n = len(co)
    res = {k: 0 for k in ("quad_iters", "quad_chunks", "quad_lanes_exp", "blocks_ideal_iters", "blocks_aabb_iters", "wide_iters",
                          "wide_chunks", "blk_pairs_ideal", "blk_pairs_aabb", "quad_staged")}
    if n == 0:
        return res
    px = x0 + np.arange(16)
    py = y0 + np.arange(16)
    dx = uv[:, 0][:, None, None] - px[None, None, :]
    dy = uv[:, 1][:, None, None] - py[None, :, None]
    power = -0.5 * (co[:, 0][:, None, None] * dx * dx + co[:, 2][:, None, None] * dy * dy) - co[:, 1][:, None, None] * dx * dy
    passes = (power <= 0) & (power >= cut[:, None, None])                       # [n,16,16] render.comp:68,78
    alpha = np.minimum(0.99, co[:, 3][:, None, None] * np.exp(np.minimum(power, 0)))
    inside = (py[:, None] < h) & (px[None, :] < w)
    # the T chain per pixel: alive_before[e] = the pixel has not broken before entry e
    one_minus = np.where(passes, 1 - alpha, 1.0)
    T = np.cumprod(one_minus, axis=0)
    broke = passes & (T < 1e-4)
    dead_after = np.cumsum(broke, axis=0) > 0
    alive_before = np.concatenate([np.ones((1, 16, 16), bool), ~dead_after[:-1]]) & inside[None]
    # alpha-cut ellipse bounding box (half extents), for the aabb variant
    det = co[:, 0] * co[:, 2] - co[:, 1] ** 2
    tau = -cut
    with np.errstate(invalid="ignore", divide="ignore"):
        hx = np.sqrt(np.maximum(2 * tau * co[:, 2] / det, 0)) + 1e-3
        hy = np.sqrt(np.maximum(2 * tau * co[:, 0] / det, 0)) + 1e-3
    keepable = cut <= 0
    for qy in (0, 8):
        for qx in (0, 8):
            sl = (slice(None), slice(qy, qy + 8), slice(qx, qx + 8))
            al = alive_before[sl]
            q_alive_any = al.reshape(n, -1).any(axis=1)
            if not q_alive_any[0]:
                continue
            last = n if q_alive_any.all() else int(np.argmin(q_alive_any))  # entries [0, last) are visited by this wave
            n_chunks = (last + 63) // 64
            # the wave stops at the chunk in which its last pixel dies (it finishes that chunk's classification, not its pairs)
            mq = min_q_rect(co[:, 0], co[:, 1], co[:, 2], uv[:, 0], uv[:, 1], x0 + qx, x0 + qx + 7.0, y0 + qy, y0 + qy + 7.0)
            kept = keepable & ~(mq > tau)
            kept[last:] = False
            res["quad_chunks"] += n_chunks
            res["quad_staged"] += min(n, n_chunks * 64)
            res["quad_iters"] += int(kept.sum())
            res["quad_lanes_exp"] += int((passes[sl] & al)[kept].sum())
            # per-block streams
            ideal = np.zeros((n, 4), bool)
            aabb = np.zeros((n, 4), bool)
            k = 0
            for by in (0, 4):
                for bx in (0, 4):
                    bs = (slice(None), slice(qy + by, qy + by + 4), slice(qx + bx, qx + bx + 4))
                    blk_alive = alive_before[bs].reshape(n, -1).any(axis=1)
                    ideal[:, k] = kept & (passes[bs] & alive_before[bs]).reshape(n, -1).any(axis=1)
                    xa, ya = x0 + qx + bx, y0 + qy + by
                    hit = (uv[:, 0] + hx >= xa) & (uv[:, 0] - hx <= xa + 3) & (uv[:, 1] + hy >= ya) & (uv[:, 1] - hy <= ya + 3)
                    aabb[:, k] = kept & hit & blk_alive
                    k += 1
            res["blk_pairs_ideal"] += int(ideal.sum())
            res["blk_pairs_aabb"] += int(aabb.sum())
            for c in range(n_chunks):
                cs = slice(64 * c, 64 * c + 64)
                res["blocks_ideal_iters"] += int(ideal[cs].sum(axis=0).max())
                res["blocks_aabb_iters"] += int(aabb[cs].sum(axis=0).max())
    for qy in (0, 8):  # two pixels per lane: 16x8 per wave
        sl = (slice(None), slice(qy, qy + 8), slice(None))
        al = alive_before[sl]
        alive_any = al.reshape(n, -1).any(axis=1)
        if not alive_any[0]:
            continue
        last = n if alive_any.all() else int(np.argmin(alive_any))
        mq = min_q_rect(co[:, 0], co[:, 1], co[:, 2], uv[:, 0], uv[:, 1], x0, x0 + 15.0, y0 + qy, y0 + qy + 7.0)
        kept = keepable & ~(mq > tau)
        kept[last:] = False
        res["wide_iters"] += int(kept.sum())
        res["wide_chunks"] += (last + 63) // 64
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B")
    ap.add_argument("--tiles", type=int, default=160)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    n, w, h, kind = WORKLOADS[args.workload]
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    verts = oracle.activate_records(pkg.synth.synth_records(n, seed=0, kind=kind))
    u = oracle.camera_uniforms(oracle.default_camera(), w, h)
    st = oracle.stages(verts, u)
    attr, bounds, payload = st["attr"], st["boundaries"], st["sorted_payload"]
    tx, ty = (w + 15) // 16, (h + 15) // 16
    rng = np.random.default_rng(1)
    tiles = rng.choice(tx * ty, size=min(args.tiles, tx * ty), replace=False)
    opac = attr["conic_opacity"][:, 3].astype(np.float32)
    cut_all = oracle.alpha_cut(opac)
    total = None
    entries = 0
    for t in tiles:
        a, b = int(bounds[2 * t]), int(bounds[2 * t + 1])
        ids = payload[a:b]
        entries += len(ids)
        r = simulate_tile(attr["conic_opacity"][ids].astype(np.float64), attr["uv"][ids].astype(np.float64),
                          cut_all[ids].astype(np.float64), (t % tx) * 16, (t // tx) * 16, w, h)
        total = r if total is None else {k: total[k] + r[k] for k in r}
    scale = tx * ty / len(tiles)
    out = {"workload": f"{kind}({n})@{w}x{h}", "tiles_sampled": int(len(tiles)), "entries_sampled": entries,
           "scaled_to_frame": {k: int(v * scale) for k, v in total.items()}}
    q = total["quad_iters"]
    out["relative_to_quad_iterations"] = {
        "blocks_ideal": round(total["blocks_ideal_iters"] / q, 4), "blocks_aabb": round(total["blocks_aabb_iters"] / q, 4),
        "wide_iters_x2 (two evaluations per iteration)": round(2 * total["wide_iters"] / q, 4),
        "quad_lane_utilisation (lanes reaching exp / 64 / iterations)": round(total["quad_lanes_exp"] / 64 / q, 4),
        "blocks_ideal_lane_pairs / quad lane pairs": round(total["blk_pairs_ideal"] * 16 / (q * 64), 4),
        "blocks_aabb_lane_pairs / quad lane pairs": round(total["blk_pairs_aabb"] * 16 / (q * 64), 4)}
    print(json.dumps(out, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
