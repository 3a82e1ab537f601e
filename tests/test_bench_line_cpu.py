"""bench.py's stdout contract: the LAST line is the one the driver parses; it holds the contract's fields only and
stays under 1 800 characters (the driver keeps a stdout tail of a few KB: round 4's 25 KB line arrived without its head
and was not parsed).  Every leg is its own short line printed before it.  Canned records: the full runs of rounds 4 and 5."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import pytest  # noqa: E402

RECORDS = [os.path.join(ROOT, "profiles", "r04", "bench_gpus1_with_legs.json"), os.path.join(ROOT, "profiles", "r05", "bench_legs.json"),
           os.path.join(ROOT, "profiles", "r06", "bench_legs.json")]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.fixture(params=RECORDS, ids=["r04", "r05", "r06"])
def canned(request):
    with open(request.param) as f:
        return json.load(f)


def test_headline_line_is_short_and_round_trips(canned):
    out = canned
    s = bench.compact_line(out, 6)
    assert len(s) < 1800 and "\n" not in s
    d = json.loads(s)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind")) <= set(d["cpu_baseline"])
    assert len(d["config"]["arithmetic"]) <= 80 and "model" not in d["config"]
    assert d["roofline"]["frac"] == out["roofline"]["frac"]


def test_eight_rank_line_is_short(canned):
    out = canned
    out["n_gpus"] = 8
    out["config"]["per_rank"] = [{"rank": r, "ms_per_step_incl_allreduce": 27.1234, "allreduce_ms_total": 1.234,
                                  "allreduces": 3} for r in range(8)]
    out["cpu_baseline"] = None
    s = bench.compact_line(out, 6)
    assert len(s) < 1800
    assert len(json.loads(s)["config"]["per_rank_ms_per_step"]) == 8


def test_emit_prints_legs_first_and_headline_last(tmp_path, canned):
    out = canned
    out["_products"] = 6
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(out, str(tmp_path / "bench_legs.json"))
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 1 + len(out["legs"])
    for ln in lines[:-1]:
        d = json.loads(ln)
        assert "leg" in d and len(ln) <= 1500
    last = json.loads(lines[-1])
    assert last["metric"] == bench.METRIC and "legs" not in last
    full = json.load(open(tmp_path / "bench_legs.json"))
    assert set(full["legs"]) == set(out["legs"])            # nothing is lost: the full record is on disk
    # the driver's tail (a few KB) always holds the whole last line
    assert len(lines[-1]) < 1800 < 8000


def test_failed_leg_is_reported_short():
    s = bench.compact_leg("x", {"error": "RuntimeError: " + "y" * 5000, "leg_wall_s": 0.1})
    assert len(s) < 1500 and json.loads(s)["leg"] == "x"


def test_oversized_fields_are_truncated_not_fatal(canned):
    """a long workload string / error / many ranks must shorten the line, never abort the run before it is printed."""
    out = canned
    out["config"]["workload"] = "w" * 5000
    out["config"]["stream_mode"] = "s" * 3000
    out["config"]["per_rank"] = [{"rank": r, "ms_per_step_incl_allreduce": 27.123456789, "allreduce_ms_total": 1.23456789,
                                  "allreduces": 3} for r in range(64)]
    out["roofline"]["kernel"] = "k" * 4000
    s = bench.compact_line(out, 6)
    assert len(s) <= 1800
    d = json.loads(s)
    assert d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"] and d["roofline"]["frac"] == out["roofline"]["frac"]
    leg = {"value": 1.0, "unit": "u", "config": {"workload": "w" * 9000}, "roofline": {"kernel": "k" * 9000, "frac": 0.5},
           "cpu_baseline": {"value": 1, "unit": "u", "cores": 1, "kind": "port", "sample": "s" * 9000}}
    s = bench.compact_leg("big", leg)
    assert len(s) <= 1500 and json.loads(s)["value"] == 1.0
