"""bench.py carries a Python restatement of include/spmv_synth.h for its
in-bench result check (the bench may not call the oracle outside its
cpu_baseline leg); it must agree with the C definition bit for bit."""
import importlib.util
import os

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location(
        "bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_python_generator_port_equals_c_definition():
    b = load_bench()
    for kind, N, K, W in ((1, 100_000, 32, 200_000), (1, 100_000, 32, 4096),
                          (1, 80_000_000, 32, 160_000_000), (0, 5000, 16, 0)):
        for g in (0, 1, 777, N // 2, N - 1):
            got = b.synth_row_dot(kind, N, K, W, 42, 7, g)
            want = O.synth_row_dot(kind, N, N, K, W, 0, 42, 7, g)
            assert got[0] == want[0], (kind, N, g)
            assert abs(got[1] - want[1]) <= 1e-15 * max(1.0, want[1])
    assert b.synth_row_dot(2, 1000, 32, 64, 42, 7, 5) is None


def test_traffic_lookup_matches_only_same_workload_and_kernel():
    b = load_bench()
    assert b.measured_traffic("no such workload", "hll_tile_panels") is None
