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


def test_traffic_of_a_templated_kernel_is_that_of_its_most_launched_shape(tmp_path, monkeypatch):
    import json
    b = _bench()
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps({"k<256>": {"launches": 70, "hbm_bytes": 5.0e7}, "k<1024>": {"launches": 3, "hbm_bytes": 4.0e7},
                                "plain": {"launches": 1, "hbm_bytes": 1.0e6}}))
    monkeypatch.setattr(b, "TRAFFIC_FILE", str(path))
    assert b.measured_traffic("k") == 5.0e7 and b.measured_traffic("plain") == 1.0e6 and b.measured_traffic("kk") is None


def test_committed_instruction_counts_are_readable():
    b = _bench()
    one = b.valu_issue(["resolve_surface_fast_kernel"], 0.037)
    frame = b.valu_issue(["setup_kernel", "raster_kernel", "resolve_surface_fast_kernel"], 0.063)
    assert one is not None and frame is not None
    assert 0.2 < one["frac"] < 1.0 and 0.2 < frame["frac"] < 1.0
    assert frame["wave_instructions"] > one["wave_instructions"] > 1e6
    # frac = instructions x mean issue cycles of the kernel's instruction mix / (1024 SIMDs x time x 2.4 GHz)
    assert 2.3 <= one["issue_cycles_per_instruction"] <= 4.0
    assert abs(one["frac"] - one["wave_instructions"] * one["issue_cycles_per_instruction"] / (1024 * 0.037e-3 * 2.4e9)) < 1e-9
    # the other two units a gather-heavy kernel can saturate: L1 tag lookups and the texture-address units, per compute unit
    assert 0.05 < one["l1_tag_lookup_frac"] < 1.0 and 0.05 < one["texture_address_unit_frac"] < 1.0
    assert b.valu_issue(["no_such_kernel"], 0.05) is None


def test_gpus_flag_starts_the_ranks_or_refuses_a_mismatch():
    """`bench.py --gpus 2` outside a launcher must not quietly run one rank and report n_gpus 1."""
    b = _bench()
    assert b.launch_plan(b.parse_args([]), {}) == ("run", 1)
    assert b.launch_plan(b.parse_args(["--gpus", "2"]), {}) == ("spawn", 2)
    assert b.launch_plan(b.parse_args(["--gpus", "8"]), {"WORLD_SIZE": "8"}) == ("run", 8)
    assert b.launch_plan(b.parse_args(["--gpus", "1"]), {"WORLD_SIZE": "1"}) == ("run", 1)     # a one-rank rehearsal under the launcher
    kind, text = b.launch_plan(b.parse_args(["--gpus", "4"]), {"WORLD_SIZE": "2"})
    assert kind == "error" and "WORLD_SIZE=2" in text
    kind, _ = b.launch_plan(b.parse_args([]), {"WORLD_SIZE": "2"})
    assert kind == "error"


def test_gpus_flag_without_a_launcher_does_not_report_one_gpu(tmp_path):
    """End to end on a machine without a GPU: the two ranks are started (and fail for want of a GPU); what must not
    happen is a successful one-rank run."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("on a GPU box this would run the whole benchmark")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-prefilter", "--no-cpu-baseline"],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert proc.returncode != 0
    assert '"n_gpus": 1' not in proc.stdout
    assert "needs a GPU" in proc.stdout


def test_modes_and_their_defaults():
    b = _bench()
    a = b.parse_args([])
    assert (a.mode, a.width, a.height, a.steps, a.warmup, a.in_flight, a.views) == ("orbit", 3840, 2160, 360, 20, 3, 4)
    assert (a.steps % a.views, a.warmup % a.views) == (0, 0)   # whole passes in the timed region and before it
    s = b.parse_args(["--mode", "strips"])
    assert (s.width, s.height, s.steps) == (7680, 4320, 20)


def test_result_names_what_was_measured():
    """The JSON line carries the fields a reader needs to tell the isolated kernel time from the in-flight one, the
    nominal bytes from the measured traffic, and the single-frame latency from the pipelined rate."""
    text = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"single_frame_ms"', '"bound_measured"', '"launch_ms"', '"in_flight"', '"traffic_frac"', '"executed_samples"', '"reference_samples"',
                '"cpu_baseline"', '"roofline"', '"frame_roofline"', '"views_per_pass"', '"views_per_launch"'):
        assert key in text, key
    assert "check_lanes(lanes)" in text          # every context is synchronised through the library after the timed region
    b = _bench()
    a = b.parse_args([])                         # the committed table is that of the default run: whole passes
    frame, launch = b.measured_frame_traffic(a.views, a.in_flight), b.measured_traffic(b.surface_shape(a.in_flight, a.views))
    assert frame is not None and launch is not None and frame > launch / a.views   # a frame is more than its share of the surface launch


def test_prefilter_roofline_reproduces_from_the_committed_counters():
    """`ibl_prefilter.roofline`: the fraction is executed samples x 110 flop over the time against the 157.3 TFLOP/s vector
    peak, and its two utilisation figures are derived from the committed counter summary exactly this way."""
    b = _bench()
    c = b.prefilter_counters()
    assert c is not None
    table = {}
    with open(b.PREFILTER_COUNTER_FILE) as f:
        shape = None
        for line in f:
            if " dispatches " in line:
                shape = "1024>" in line or "prefilter_levels_kernel" in line
            elif shape and len(line.split()) >= 2 and line.split()[0][:1].isupper() and "_" in line.split()[0]:   # (counter lines; the summary ends with kernel-trace rows)
                table.setdefault(line.split()[0], float(line.split()[1]))
    cycles = table["GRBM_GUI_ACTIVE"] / 8                       # the profiler sums the eight XCDs' counters
    assert abs(c["ta_busy_frac"] - table["TA_TA_BUSY_sum"] / 256 / cycles) < 1e-12
    assert abs(c["valu_issue_frac"] - table["SQ_INSTS_VALU"] * b.PREFILTER_ISSUE_CYCLES_PER_INSTRUCTION / (1024 * cycles)) < 1e-12 and 2.3 <= b.PREFILTER_ISSUE_CYCLES_PER_INSTRUCTION <= 4.0
    assert 0.5 < c["ta_busy_frac"] < 1.0 and 0.6 < c["valu_issue_frac"] <= 1.05
    r = b.prefilter_roofline(8_600_000_000, 15.5, c)
    assert r["bound"] == "fp32_valu" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - 8.6e9 * 110 / 15.5e-3 / 1e12 / 157.3) < 1e-12 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-12
    assert r["valu_issue"] == c["valu_issue_frac"] and r["ta_busy"] == c["ta_busy_frac"]
    assert b.prefilter_counters("/nonexistent") is None


def test_result_has_a_roofline_for_each_dominant_kernel_and_each_headline():
    text = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"roofline_raster"', '"kernel_ms_alone_per_pass"', '"roofline": prefilter_roofline(', '"exchange_bytes_per_rank"', '"replicated_texels"'):
        assert key in text, key
    b = _bench()
    assert b.measured_traffic("raster_views_kernel") is not None      # the raster entry's `traffic`
    assert b.valu_issue(["raster_kernel"], 0.025) is not None         # and its `valu_issue`


def test_passes_cover_every_step_exactly_once():
    """Whatever the step count, the views per pass and the contexts in flight: the passes submitted over the steps of a
    region draw each of its frames once, none twice, none of another region; contexts take the passes in turn."""
    b = _bench()
    for count in (0, 1, 2, 5, 7, 20, 360, 361):
        for views in (1, 2, 3, 4):
            for contexts in (1, 3, 4):
                drawn, lanes_used = [], []
                for i in range(count):
                    sub = b.pass_of_step(i, count, views, contexts)
                    if sub is None:
                        continue
                    lane, lo, hi = sub
                    assert lo == i and lo < hi <= count and hi - lo <= views and 0 <= lane < contexts
                    drawn.extend(range(lo, hi))
                    lanes_used.append(lane)
                assert drawn == list(range(count)), (count, views, contexts)
                assert lanes_used == [k % contexts for k in range(len(lanes_used))]
