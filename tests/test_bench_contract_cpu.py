"""The committed bench records carry every field of the bench.py contract (one JSON line per run)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _records():
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json"))):
        line = open(path).read().strip().splitlines()[-1]
        yield path, json.loads(line)


def test_committed_bench_lines_follow_the_contract():
    seen = 0
    for path, d in _records():
        seen += 1
        for k in REQUIRED:
            assert k in d, (path, k)
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
        assert d["dtype"] == "f32" and d["vs_baseline"] is None
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (path, k)
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        c = d["cpu_baseline"]
        if c is not None:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in c, (path, k)
            assert c["kind"] in ("port", "reference")
        assert abs(d["value"] - d["config"]["channels_per_gpu"] * d["n_gpus"] * d["config"]["frames_per_block"]
                   / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert seen >= 4


def test_headline_record_names_the_baseline_metric():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    d = dict(_records())[os.path.join(ROOT, "profiles", "r01_bench_1gpu.json")]
    assert d["unit"] == "samples/s" and "samples/sec" in d["metric"]
    assert d["roofline"]["traffic"] and 0.9 < d["roofline"]["traffic"] / (16.5 * (1 << 20) * 128) < 1.1
    assert isinstance(base, dict)


def test_the_printed_line_fits_the_drivers_stdout_tail():
    """The driver keeps the last 8 KB of stdout (BENCH_r04: the line was 16 KB and cfg3 fell off the front).  bench.py prints
    compact_line(full record): every contract field unchanged, every config's numbers, under 7 KB; the full record goes to
    stderr and gpurun_out/."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = 0
    for path, d in _records():
        if "other_configs" not in d:
            continue
        seen += 1
        c = bench.compact_line(d)
        text = json.dumps(c)
        assert len(text) < 7000, (path, len(text))
        for k in REQUIRED:
            assert k in c, (path, k)
            if not isinstance(d[k], (dict, list)):
                assert c[k] == d[k], (path, k)
        assert c["roofline"]["frac"] == pytest_approx(d["roofline"]["frac"]) and c["roofline"]["bound"] == d["roofline"]["bound"]
        assert set(c["other_configs"]) == set(d["other_configs"])
        for name, o in d["other_configs"].items():
            if "error" not in o:
                assert c["other_configs"][name]["roofline"]["kernel"] == o["roofline"]["kernel"]
    assert seen >= 1


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-6)


def test_round6_records_carry_the_host_boundary_and_say_which_cpu_build_ran():
    """the driver-shaped line reports `host_path` (dspfx_process_host from page-locked buffers: what
    GpuBank::process costs, PCIe-bound, never `value`) and `cpu_baseline.build` (rebuilt -O3 -march=native on the box, or the
    prebuilt fallback); the self-launched N = 2 record reports all three forms of the bus exchange."""
    recs = {os.path.basename(p): d for p, d in _records()}
    d = recs["r06_bench_driver_cmd_run1.json"]
    h = d["host_path"]
    assert h["cfg5_shard"]["channels"] == 1 << 20 and h["cfg5_shard"]["ms_per_block_p50"] > d["ms_per_step"] * 5      # PCIe-bound: far from `value`
    assert h["cfg5_shard"]["inside_budget_p99"] is False and h["largest_realtime_pow2"]["inside_budget_p99"] is True
    assert h["largest_realtime_pow2"]["ms_per_block_p99"] < 128 / 48.0
    assert d["cpu_baseline"]["build"].startswith("gcc -O3 -march=native")
    two = json.loads(open(os.path.join(ROOT, "profiles", "r06_two_ranks_self_launched.json")).read().strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["cpu_baseline"] is None and two["collective_backend"] == "mailbox"
    f = two["scaling_forms"]
    assert abs(f["inline"]["value"] - two["value"]) / two["value"] < 1e-6 and f["inline"]["bus_delay_blocks"] == 0       # (the compact line rounds nested floats to 7 digits)
    assert f["same_block_second_stream"]["bus_delay_blocks"] == 0 and f["overlapped"]["bus_delay_blocks"] == 2
    assert all(f[k]["value"] > 0 for k in f)
