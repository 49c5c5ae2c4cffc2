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
    return entry.load_oracle()


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
