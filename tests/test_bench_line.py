"""The ONE stdout line of bench.py (a 23.9 KB line left BENCH_r05.json unparsed): built from canned records, it must be
short, strict JSON, and carry the contract's fields plus `roofline` and `cpu_baseline`."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fail(c):
    raise AssertionError("non-JSON constant %r in the bench line" % c)


CANNED = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_final_bench.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]_final_bench_records.json")))
# (since round 6 `rNN_final_bench.json` IS the compact line bench.py printed; the full record it was cut from is `rNN_final_bench_records.json`)
FULL = [p for p in CANNED if "parity" in json.load(open(p)).get("config", {}) and "residual" in json.load(open(p))]


@pytest.mark.parametrize("path", CANNED, ids=[os.path.basename(p) for p in CANNED])
def test_line_from_a_committed_full_record_is_short_strict_json(path):
    m = _bench()
    out = json.load(open(path))
    if "roofline" not in out:          # (a committed compact line: nothing to rebuild it from)
        pytest.skip("compact record")
    txt = m.compact_line(out, "bench_records.json")
    assert "\n" not in txt and len(txt) < 6000, len(txt)
    assert len(txt) <= m.LINE_BYTES_MAX
    line = json.loads(txt, parse_constant=_fail)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(out["value"], rel=1e-6)
    assert "workload" in line["config"] and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(out["roofline"]["frac"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k


def test_non_finite_numbers_and_oversized_records_do_not_break_the_line():
    m = _bench()
    out = json.load(open(FULL[-1]))
    out["roofline"]["traffic"] = float("nan")
    out["residual"]["ipm_max_gap"] = float("inf")
    out["value_to_convergence"] = float("nan")
    out["config"]["parity"]["teacher_forced"] = {"case_%03d" % i: {"optimal_value_rel_diff_max": 1e-7} for i in range(400)}
    txt = m.compact_line(out, None)
    assert len(txt) <= m.LINE_BYTES_MAX
    line = json.loads(txt, parse_constant=_fail)
    assert line["roofline"]["traffic"] is None and "cpu_baseline" in line and "roofline" in line
    # the side file's cleaner maps non-finite numbers to null too
    import numpy as np
    rec = {"a": np.float64("nan"), "b": [np.int64(3), float("inf")], "c": np.arange(3)}
    import tempfile
    old = m.ROOT
    with tempfile.TemporaryDirectory() as d:
        m.ROOT = d
        try:
            assert m.write_records(rec) == "bench_records.json"
            assert json.load(open(os.path.join(d, "bench_records.json")), parse_constant=_fail) == {"a": None, "b": [3, None], "c": [0, 1, 2]}
        finally:
            m.ROOT = old
