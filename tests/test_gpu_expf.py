"""The blend's exp() ON THE DEVICE, exhaustively (render.comp:77).

  * exp mode 2's gs_expf_libm (csrc/gs_device.h: glibc's expf restated in binary64, the four-operation cubic, v_cvt at the end,
    plus libm's underflow to 0) is evaluated by a kernel on EVERY binary32 <= 0 -- 2 139 095 041 values -- and compared with this
    machine's libm through per-block checksums (gs_debug_expf_scan / oracle.libm_expf_block_sums; a mismatching block is
    re-compared value by value).  tests/test_expf_libm.py proves the CPU restatement of the same operation sequence; this is the
    device code itself -- the inline v_fma_f64, the table, the conversions.
  * exp mode 3's guard rests on |v_exp_f32(fl(x log2e)) - expf(x)| <= (E0 - 2^-23 + E1 |x|) expf(x): measured over every binary32
    in [-16, 0] by the same kernel and asserted against the constants the kernel is built with.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIRST, LAST = 0x80000000, 0xFF800000  # -0.0 .. -inf


def test_gs_expf_libm_on_the_device_equals_libm_on_every_nonpositive_binary32(pkg, oracle, gpu, _sort_path):
    if _sort_path == "1":
        pytest.skip("independent of the depth-order path: runs once")
    count = LAST - FIRST + 1
    dev, guard = pkg.debug_expf_scan(FIRST, count)
    host = oracle.libm_expf_block_sums(FIRST, count)
    bad = np.nonzero(dev != host)[0]
    assert len(bad) == 0, f"{len(bad)} of {len(dev)} blocks of 2^20 values differ from libm's expf, first block {bad[0]} (bits {FIRST + (int(bad[0]) << 20):#x} ..)"
    # the checker's restatement agrees with the same libm (the CPU suite's exhaustive pin, repeated here on the GPU box's libm)
    n_bad, first = oracle.expf_libm_mismatches(FIRST, count)
    assert n_bad == 0, (n_bad, hex(first))
    assert oracle.expf_monotone_violations(FIRST, count - 1) == 0  # the premise of the alpha cut, on this box's libm
    print(f"gs_expf_libm on the device == libm expf on all {count} binary32 <= 0 ({len(dev)} block checksums)")


def test_the_guards_premise_holds_for_v_exp_f32(pkg, oracle, gpu, _sort_path):
    if _sort_path == "1":
        pytest.skip("independent of the depth-order path: runs once")
    lo = int(np.float32(-16.0).view(np.uint32))
    _, guard = pkg.debug_expf_scan(FIRST, lo - FIRST + 1)
    worst, near, e0, e1 = (float(g) for g in guard)
    print(f"v_exp_f32(fl(x log2e)) vs expf(x): max(rel - {e1:.3g} |x|) = {worst:.4g} over [-16, 0], max rel over [-1, 0] = {near:.4g}; "
          f"guard E0 = {e0:.4g} (needs >= {worst + 2.0 ** -23:.4g}), E1 = {e1:.3g}")
    assert worst + 2.0 ** -23 <= e0, (worst, e0)
