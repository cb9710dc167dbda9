import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "candle-video_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file

    def load(name):
        return load_file(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session", autouse=True)
def _build_library():
    """The C-ABI library must exist for both suites (CPU suite checks that it loads and exports)."""
    so = os.path.join(ROOT, "candle-video_amd", "libltxhip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    yield


@pytest.fixture(autouse=True)
def _default_options():
    """Run-time options (ltx_set_option, include/ltxhip.h) set by a test never leak into the next one."""
    yield
    m = sys.modules.get("ltxhip")
    if m is not None:
        m.reset_options()


def rel_max(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
