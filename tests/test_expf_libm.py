"""Pins gso_expf_libm -- glibc's expf algorithm restated in binary64 (oracle/gs_oracle.c; the HIP blend's gs_expf_libm
performs the same ten binary64 operations) -- to THIS MACHINE'S libm, exhaustively.

Why it matters: the reference's shader text compiled for the CPU (oracle/_ref) evaluates render.comp:77's exp() with
libm's expf.  With exp() identical on every input the blend can be, and is, bit-identical to the reference text
(tests/test_oracle_vs_ref.py on the CPU, tests/test_gpu_blend_modes.py and test_gpu_full_size.py on the GPU).

glibc's expf (sysdeps/ieee754/flt-32/e_expf.c since 2.27) is third-party code absent from /root/reference: the
algorithm is restated from its published form (ARM optimized-routines, 2017), the 32-entry table is GENERATED below and
compared with the one compiled into the oracle, and the result is checked against libm on every binary32 the blend
can ask for (power <= 0: 2 139 095 041 values, a few seconds with OpenMP).
"""
import ctypes as C
import struct

import numpy as np


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_equal_to_libm_on_every_nonpositive_binary32(oracle):
    lo, hi = _bits(-0.0), _bits(float("-inf"))
    bad, first = oracle.expf_libm_mismatches(lo, hi - lo + 1)
    assert bad == 0, f"{bad} mismatches against libm expf, first at bits {first:#x}"


def test_the_kernels_operation_sequence_equals_libm_too(oracle):
    """The HIP blend evaluates the same cubic times the same table value in four binary64 operations instead of glibc's
    five (gs_expf_libm in gs_blend.hip; gso_expf_device restates exactly that sequence).  IEEE binary64 arithmetic
    gives the same bits on the host as on the device, so its equality with libm on every input the blend can ask for is
    decided here, exhaustively -- and re-checked end to end by the GPU tests, where the frame must equal the reference
    text's bit for bit."""
    lo, hi = _bits(-0.0), _bits(float("-inf"))
    bad, first = oracle.expf_device_mismatches(lo, hi - lo + 1)
    assert bad == 0, f"{bad} mismatches against libm expf, first at bits {first:#x}"


def test_libm_expf_is_monotone_on_every_nonpositive_binary32(oracle):
    """The premise of the ALPHA CUT (csrc/gs_device.h: alpha_cut; oracle: gso_alpha_cut): expf never grows as x falls, so
    render.comp:78's `alpha < 1/255` is a threshold on `power` -- checked on every adjacent pair of binary32 values <= 0."""
    lo, hi = _bits(-0.0), _bits(float("-inf"))
    assert oracle.expf_monotone_violations(lo, hi - lo) == 0


def test_alpha_cut_is_the_threshold_of_render_comp_78(oracle):
    """gso_alpha_cut(o) = the most negative power kept: kept at the cut, not kept one binary32 below, for a sweep of opacities
    (sigmoid range, the neighbourhood of 1/255, out-of-range values) -- evaluated with the literal :77-78 on libm's expf."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(0)
    third = np.float32(1 / 255)
    ops = np.concatenate([1 / (1 + np.exp(-rng.normal(0, 2.5, 3000))), rng.uniform(0.0039, 0.0041, 500), rng.uniform(0.9, 1.0, 300),
                          [0.0, -1.0, third, np.nextafter(third, np.float32(0)), np.nextafter(third, np.float32(1)), 1.0, 2.0, 1e30,
                           np.inf, np.nan, 1e-40]]).astype(np.float32)
    cut = oracle.alpha_cut(ops)

    def kept(o, p):
        a = np.float32(o) * np.float32(libm.expf(float(p)))
        a = np.float32(0.99) if np.isnan(a) else min(np.float32(0.99), a)
        return not a < third
    with np.errstate(all="ignore"):
        for o, c in zip(ops, cut):
            if np.isposinf(c):
                assert not kept(o, np.float32(-0.0)), (o, c)
            elif np.isneginf(c):
                assert kept(o, np.float32(-np.inf)) and kept(o, np.float32(-80.0)), (o, c)
            else:
                assert c <= 0 and kept(o, c) and not kept(o, np.nextafter(np.float32(c), np.float32(-np.inf))), (o, c)
                # and monotone in between: a few powers above the cut are kept, a few below are not
                for p in (c * 0.5, c * 0.999, np.float32(-0.0)):
                    assert kept(o, np.float32(p)), (o, c, p)
                for p in (c * 1.001 - 1e-6, c - 1.0):
                    assert not kept(o, np.float32(p)), (o, c, p)
    assert cut[ops == 1.0][0] == np.float32(-5.541263)  # ln(1/255) = -5.5413: the whole sigmoid range lies above -5.55


def test_spot_values_and_table(oracle):
    assert oracle.expf_libm(np.float32(0.0)) == 1.0 and oracle.expf_libm(np.float32(-0.0)) == 1.0
    assert oracle.expf_libm(np.float32(-1000.0)) == 0.0 and oracle.expf_libm(np.float32(-np.inf)) == 0.0
    x = np.linspace(-8, 0, 4001).astype(np.float32)
    e = oracle.expf_libm(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    ulp = np.abs(e - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 0.502  # glibc's stated bound; correctly rounded in all but ~0.2 % of the cases
    # the table: 2^(i/32) correctly rounded to binary64, with i << 47 subtracted from the bit pattern.  exp2(i/32) in
    # binary64 (numpy -> libm exp2, itself < 1 ULP) may be off by one in the last place: recompute exactly with fractions
    from fractions import Fraction
    lib = oracle.lib()
    tab = (C.c_uint64 * 32).in_dll(lib, "k_expf_tab_export")
    for i in range(32):
        # correctly rounded 2^(i/32): find the binary64 y with y^32 closest to 2^i by exact rational comparison
        approx = float(2.0 ** (i / 32.0))
        cands = [np.nextafter(approx, 0), approx, np.nextafter(approx, 4)]
        target = Fraction(2) ** i

        def err(c):  # |c - 2^(i/32)| is monotone in |c^32 - 2^i| near the root
            return abs(Fraction(float(c)) ** 32 - target)
        best = min(cands, key=err)
        # the rounding boundary: best must beat the midpoints to its neighbours
        for nb in (np.nextafter(best, 0), np.nextafter(best, 4)):
            mid = (Fraction(float(best)) + Fraction(float(nb))) / 2
            assert (mid ** 32 < target) == (nb < best), i
        bits = struct.unpack("<Q", struct.pack("<d", float(best)))[0]
        assert tab[i] == (bits - (i << 47)) & 0xFFFFFFFFFFFFFFFF, i
