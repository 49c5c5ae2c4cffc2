"""Synthetic scene S(N) and PLY I/O in the reference's on-disk format.

A scene is an (N, 62) float32 array of PLY-domain records, the byte layout GSScene::load reads
(/root/reference/src/GSScene.cpp:17-24): x y z | nx ny nz | f_dc 0..2 | f_rest 0..44 (planar: 15 R,
15 G, 15 B) | opacity (logit) | scale 0..2 (log) | rot 0..3 (w x y z, un-normalised).

The generator is counter based (PCG hash of seed, Gaussian index and slot) so any slice of a scene
can be produced independently and reproducibly; pure numpy, used by tests and bench.py.
"""
import numpy as np

RECORD_FLOATS = 62
_SLOTS = 128  # counter stride per Gaussian


def _pcg_hash(x):
    x = x.astype(np.uint32)
    state = x * np.uint32(747796405) + np.uint32(2891336453)
    word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def _uniform(seed, idx, slot):
    """U[0,1) with 24 random bits, float64."""
    with np.errstate(over="ignore"):
        ctr = (np.uint32(seed) * np.uint32(0x9E3779B9) + idx.astype(np.uint32) * np.uint32(_SLOTS)
               + np.uint32(slot))
        h = _pcg_hash(ctr)
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)


def _normal_pair(seed, idx, slot):
    u1 = 1.0 - _uniform(seed, idx, slot)  # (0, 1]
    u2 = _uniform(seed, idx, slot + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


def scene_params(kind, n):
    """Distribution parameters for the named configs of BASELINE.md §2."""
    if kind == "A":  # 10 k / 256x256 plumbing config
        return dict(x=(-1.0, 1.0), y=(-1.0, 1.0), z=(-6.0, -2.0), log_scale_mean=-3.0)
    if kind == "T":  # trained-scene statistics (see synth_records): the mean is calibrated so that D stays near S(n)'s
        return dict(x=(-4.5, 4.5), y=(-2.6, 2.6), z=(-12.0, -2.0),
                    log_scale_mean=-5.35 - np.log(n / 1.0e6) / 3.0)
    # S(N): density-compensated splat size
    return dict(x=(-4.5, 4.5), y=(-2.6, 2.6), z=(-12.0, -2.0),
                log_scale_mean=-4.5 - np.log(n / 1.0e6) / 3.0)


# Scene kind "T": what a TRAINED scene looks like to the rasteriser, since no trained PLY ships with the reference or this
# container (BASELINE configs[2], the Mip-NeRF360 garden, is 6 M such Gaussians; GSScene.cpp:26-68 loads such files).
# Against S(N) (isotropic-ish splats, log-scale sigma 0.6, uniform positions, logit N(0, 2)):
#   * per-axis log-scales N(mu, 1.3) drawn independently, clipped at mu + 3.6, and 2 % of the Gaussians get one axis
#     stretched by e^2 more: needles and discs with axis ratios up to ~e^7 -- the regime in which `power` is a difference of
#     large cancelling terms and the contracted / uncontracted readings of render.comp:66 part ways;
#   * positions clustered: 80 % in 96 Gaussian blobs (sigma 0.15 .. 0.9, centres uniform in S's box), 20 % uniform background
#     -- per-tile list lengths vary by two orders of magnitude instead of S's near-uniform load;
#   * opacity bimodal: 55 % nearly transparent (logit N(-2.5, 1)), 45 % nearly opaque (logit N(3, 1.5));
#   * view-dependent colour a little stronger (f_rest N(0, 0.15)).
_T_CLUSTERS = 96


def _trained_like(rec, n, seed, idx, p):
    lerp = lambda lohi, u: lohi[0] + (lohi[1] - lohi[0]) * u  # noqa: E731
    cid = (np.floor(_uniform(seed, idx, 70) * _T_CLUSTERS)).astype(np.uint64)
    background = _uniform(seed, idx, 71) < 0.20
    cseed = np.uint32(seed) + np.uint32(7919)
    sig = 0.15 + 0.75 * _uniform(cseed, cid, 3) ** 2
    gx, gy = _normal_pair(seed, idx, 72)
    gz, _ = _normal_pair(seed, idx, 74)
    for k, (lohi, g) in enumerate(((p["x"], gx), (p["y"], gy), (p["z"], gz))):
        c = lerp(lohi, _uniform(cseed, cid, k))
        v = np.where(background, lerp(lohi, _uniform(seed, idx, k)), c + sig * g * (1.0 if k < 2 else 1.6))
        # keep the blobs' tails inside the box by folding them back at its walls (clipping would pile thousands of
        # Gaussians onto one exact coordinate -- for z: one exact depth, a degenerate key run no trained scene has)
        lo_w, hi_w = lohi[0] - 1.0, (-0.5 if k == 2 else lohi[1] + 1.0)
        v = np.where(v < lo_w, 2.0 * lo_w - v, v)
        v = np.where(v > hi_w, 2.0 * hi_w - v, v)
        rec[:, k] = np.clip(v, lo_w, hi_w)
    s0, s1 = _normal_pair(seed, idx, 3)
    s2, _ = _normal_pair(seed, idx, 5)
    stretch_axis = (np.floor(_uniform(seed, idx, 76) * 3)).astype(np.int64)
    stretched = _uniform(seed, idx, 77) < 0.02
    for k, sn in enumerate((s0, s1, s2)):
        ls = p["log_scale_mean"] + np.minimum(1.3 * sn, 3.6)
        rec[:, 55 + k] = ls + np.where(stretched & (stretch_axis == k), 2.0, 0.0)
    o, o2 = _normal_pair(seed, idx, 11)
    rec[:, 54] = np.where(_uniform(seed, idx, 78) < 0.55, -2.5 + o, 3.0 + 1.5 * o2)


def synth_records(n, seed=0, kind="S", start=0, **override):
    """Records [start, start+n) of scene `kind` with `n_total = override.get('n_total', n)`."""
    n_total = override.pop("n_total", n)
    p = scene_params(kind, n_total)
    p.update(override)
    idx = np.arange(start, start + n, dtype=np.uint64)
    rec = np.zeros((n, RECORD_FLOATS), np.float32)

    def lerp(lohi, u):
        return lohi[0] + (lohi[1] - lohi[0]) * u

    rec[:, 0] = lerp(p["x"], _uniform(seed, idx, 0))
    rec[:, 1] = lerp(p["y"], _uniform(seed, idx, 1))
    rec[:, 2] = lerp(p["z"], _uniform(seed, idx, 2))
    # normals (3..5) stay 0, GSScene.cpp:56-58
    s0, s1 = _normal_pair(seed, idx, 3)
    s2, _ = _normal_pair(seed, idx, 5)
    for k, s in enumerate((s0, s1, s2)):
        rec[:, 55 + k] = p["log_scale_mean"] + 0.6 * s
    q0, q1 = _normal_pair(seed, idx, 7)
    q2, q3 = _normal_pair(seed, idx, 9)
    for k, q in enumerate((q0, q1, q2, q3)):
        rec[:, 58 + k] = q
    o, _ = _normal_pair(seed, idx, 11)
    rec[:, 54] = 2.0 * o
    for k in range(3):
        rec[:, 6 + k] = -1.5 + 3.0 * _uniform(seed, idx, 13 + k)
    rest = 0.15 if kind == "T" else 0.1
    for k in range(0, 46, 2):
        a, b = _normal_pair(seed, idx, 16 + k)
        rec[:, 9 + k] = rest * a
        if k + 1 < 45:
            rec[:, 9 + k + 1] = rest * b
    if kind == "T":
        _trained_like(rec, n, seed, idx, p)
    return rec


_PROPS = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
          + [f"f_rest_{i}" for i in range(45)] + ["opacity"] + [f"scale_{i}" for i in range(3)]
          + [f"rot_{i}" for i in range(4)])


def write_ply(path, records):
    records = np.ascontiguousarray(records, "<f4").reshape(-1, RECORD_FLOATS)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(records)
    header += "".join(f"property float {p}\n" for p in _PROPS) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(records.tobytes())


def read_ply_records(path):
    """Raw records (no activation) -- mirrors loadPlyHeader's leniency: only `element vertex`."""
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline()
            if not line:
                raise RuntimeError("Could not find end of header")
            tok = line.split()
            if tok[:2] == [b"element", b"vertex"]:
                n = int(tok[2])
            if tok[:1] == [b"end_header"]:
                break
        data = np.frombuffer(f.read(n * RECORD_FLOATS * 4), "<f4")
    return data.reshape(n, RECORD_FLOATS).copy()
