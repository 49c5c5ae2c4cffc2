import os
import sys

import pytest

# The suite's image comparisons are assert_images_identical against the reference's shader text: that is the blend's exp
# mode 2 (libm's expf restated), which the renderers of these tests therefore start in.  The library's default, mode 3 (the
# guarded v_exp_f32), is selected explicitly by the tests that check it (test_gpu_guard.py, the full-size and fuzz tests).
os.environ.setdefault("GS_EXP_MODE", "2")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker.  Its exp() restates glibc's expf (x86-64 FMA build, glibc >= 2.27); the bitwise image tests compare with
    the reference text compiled against THIS HOST's libm.  On a host whose libm evaluates expf differently (no FMA, another libc)
    the two would disagree for reasons that have nothing to do with the kernels: a sample of the pin is taken here once, and
    helpers.assert_images_identical turns a bitwise image mismatch into an xfail with that explanation instead of a failure
    (tests/test_expf_libm.py still runs the exhaustive pin and fails loudly)."""
    o = entry.load_oracle()
    import helpers
    bad = 0
    for first in (0x80000000, 0xBF000000, 0xC0A00000, 0xC2A00000):  # -0.., -0.5.., -5.., -80..: 4 x 2^20 values
        n, _ = o.expf_libm_mismatches(first, 1 << 20)
        bad += n
    helpers.HOST_LIBM_MISMATCHES = bad
    return o


@pytest.fixture(scope="session")
def gpu(pkg):
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    return 0


def pytest_generate_tests(metafunc):
    """Every GPU test runs on both depth-order paths: the global depth order (GS_SORT_PATH=1, 26 kernels per frame)
    and the automatic choice (bin-local in-LDS sort, falling back when a bin does not fit)."""
    if metafunc.definition.get_closest_marker("gpu") and "_sort_path" in metafunc.fixturenames:
        metafunc.parametrize("_sort_path", ["1", "0"], ids=["global", "auto"], indirect=True)


@pytest.fixture(autouse=True)
def _sort_path(request, monkeypatch):
    param = getattr(request, "param", None)
    if param is not None:
        monkeypatch.setenv("GS_SORT_PATH", param)
    return param


def pytest_collection_modifyitems(config, items):
    """No GPU test may hold a box for long: a hung kernel must fail the test, not burn the GPU budget."""
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))
