"""pytest configuration: markers, library builds, shared fixtures."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _ensure_built():
    """Build the oracle and the product library if their .so files are absent
    (they are git-ignored; on the GPU box the prebuilt ones travel)."""
    need_oracle = not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))
    need_lib = not os.path.exists(
        os.path.join(ROOT, "spmv_scpa_amd", "lib", "libspmv_scpa_amd.so"))
    if os.environ.get("SPMV_SKIP_BUILD"):
        return
    if need_oracle or need_lib:
        subprocess.run([sys.executable, "-c",
                        "import __graft_entry__ as g; g.build()"],
                       cwd=ROOT, check=True)


_ensure_built()


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
