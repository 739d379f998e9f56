"""bench.py reads two committed measurement files (HBM-side traffic and vector-instruction counts of the frame
kernels).  A renamed or reshaped file would silently turn the corresponding JSON fields into null."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_traffic_measurement_is_readable():
    b = _bench()
    t = b.measured_traffic("resolve_surface_fast_kernel")
    assert t is not None and 1e7 < t < 1e9          # tens of megabytes per 4K launch
    assert b.measured_traffic("no_such_kernel") is None


def test_committed_instruction_counts_are_readable():
    b = _bench()
    one = b.valu_issue(["resolve_surface_fast_kernel"], 0.037)
    frame = b.valu_issue(["setup_kernel", "raster_kernel", "resolve_surface_fast_kernel"], 0.063)
    assert one is not None and frame is not None
    assert 0.2 < one["frac"] < 1.0 and 0.2 < frame["frac"] < 1.0
    assert frame["wave_instructions"] > one["wave_instructions"] > 1e6
    # frac = instructions x 4 cycles / (1024 SIMDs x time x 2.4 GHz)
    assert abs(one["frac"] - one["wave_instructions"] * 4 / (1024 * 0.037e-3 * 2.4e9)) < 1e-9
    assert b.valu_issue(["no_such_kernel"], 0.05) is None
