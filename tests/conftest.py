import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# property tests draw the same examples on every run (a round-end run must not depend on luck); explore other examples with
# NTK_HYPOTHESIS_RANDOM=1 (fresh random examples every run)
try:
    from hypothesis import settings as _hs
    _hs.register_profile("deterministic", derandomize=True, deadline=None)
    if not os.environ.get("NTK_HYPOTHESIS_RANDOM"):
        _hs.load_profile("deterministic")
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # the HIP library is built in-tree (python -c "import __graft_entry__ as g; g.build()"); a checkout without it gets it
    # compiled here once (hipcc cross-compiles gfx950 without a GPU) - there is nothing else to fall back to
    so = os.path.join(ROOT, "needletail_amd", "libneedletail_amd.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "needletail_amd", "csrc")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
