"""The line bench.py prints is what the driver parses: short, numbers and tokens only, every contract key present.
(Round 5's line was 24 kB with prose in it and the driver recorded `parsed: null`.)"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench as B
    finally:
        sys.argv = argv
    return B


@pytest.fixture(scope="module")
def canned():
    # a full record of a real run (round 5, the driver's flags): the input the compact line is built from
    with open(os.path.join(ROOT, "profiles", "r05_bench_driver_flags.json")) as fh:
        return json.load(fh)


def test_line_is_short_and_round_trips(bench, canned):
    line = bench.compact_line(canned)
    assert "\n" not in line
    assert len(line) <= 6000, len(line)
    back = json.loads(line)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == pytest.approx(canned["value"], rel=1e-3)
    assert back["ms_per_step"] == pytest.approx(canned["ms_per_step"], rel=1e-3)
    assert back["config"]["workload"].startswith("configs[2]")
    rf = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "pmc", "kernel", "algorithmic_ratio"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3)
    cb = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "parity_checked", "parity_mismatches"):
        assert k in cb, k
    assert set(back["extra_configs"]) == set(canned["extra_configs"])
    for name in ("words", "skewed", "geonames_x4", "geonames_miss"):
        x = back["extra_configs"][name]
        assert x["value"] == pytest.approx(canned["extra_configs"][name]["value"], rel=1e-3)
        assert x["parity_mismatches"] == 0


def test_line_holds_no_prose(bench, canned):
    back = json.loads(bench.compact_line(canned))

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    longest = max(strings(back), key=len)
    assert len(longest) <= 120, longest


def test_a_failed_leg_stays_visible_and_short(bench, canned):
    full = json.loads(json.dumps(canned))
    full["cpu_baseline"] = {"error": "RuntimeError('" + "x" * 5000 + "')"}
    full["extra_configs"]["skewed"] = {"error": "boom " * 2000}
    back = json.loads(bench.compact_line(full))
    assert "error" in back["cpu_baseline"] and len(back["cpu_baseline"]["error"]) <= 120
    assert "error" in back["extra_configs"]["skewed"]


def test_a_line_that_outgrows_the_limit_is_refused(bench, canned):
    full = json.loads(json.dumps(canned))
    full["config"]["workload"] = "w" * 7000
    with pytest.raises(RuntimeError):
        bench.compact_line(full)


def test_multi_rank_line_names_the_collective(bench, canned):
    full = json.loads(json.dumps(canned))
    full.pop("extra_configs"); full.pop("cpu_baseline")
    full.update(n_gpus=8, replicas=8, collective={"backend": "nccl", "world": 8, "rccl_version": "2.22.3", "distinct_devices": 8,
                                                  "op": "gather to rank 0 " * 10, "device_names": ["AMD Instinct MI355X"]},
                per_rank={"kernel_ms": [430.0] * 8, "gather_ms": [1.0] * 8}, gather_ms=1.0, gather_bytes_per_rank=124000000,
                gather_checked=2)
    back = json.loads(bench.compact_line(full))
    assert back["collective"] == {"backend": "nccl", "world": 8, "rccl_version": "2.22.3", "distinct_devices": 8}
    assert len(back["per_rank_kernel_ms"]) == 8
