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
    config._summary_lines = []  # tests append; printed after the run


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """lines a test wants in the log of a quiet (`-q`, captured) run: the perf
    floors print measured vs floor here, so the driver's log shows the margin
    whether or not a floor failed"""
    lines = getattr(config, "_summary_lines", [])
    if lines:
        terminalreporter.section("measured vs floor (tests/test_gpu_perf_floor.py)")
        for line in lines:
            terminalreporter.write_line(line)


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


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _release_device_objects_of_the_test(request):
    """GPU tests: whatever device objects a test created and did not release
    -- it failed before its release() calls, say -- are released right after
    it, in dependency order, instead of living on in the failure's traceback
    until the interpreter exits (round 2: such leftovers were finalised
    during interpreter shutdown and the process aborted).  Objects of wider
    fixtures (created before the test started) are left alone."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import spmv_scpa_amd as S
    before = {id(o) for o in S.live_objects()}
    ignored = S.ignored_releases()
    yield
    mine = [o for o in S.live_objects() if id(o) not in before]
    for rank in (2, 1, 0):
        for o in mine:
            if o._RANK == rank:
                o._release_now()
    # a release the library had to ignore (double release, stale wrapper) is a
    # bug in the caller even though it is harmless: only the tests that do it
    # on purpose may move the counter (make it observable)
    deliberate = ("release" in request.node.name or "stale" in request.node.name)
    if not deliberate:
        assert S.ignored_releases() == ignored, (
            "%s: the library ignored %d release(s) of dead handles"
            % (request.node.name, S.ignored_releases() - ignored))
