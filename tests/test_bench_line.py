"""bench.py's LAST stdout line must stay small enough for the driver to parse (a 20.7 KB line left
BENCH_r03.parsed null).  The composer is exercised on the real round-3 record and on an inflated worst case."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _real_record():
    return json.load(open(os.path.join(ROOT, "profiles", "round3", "bench_default_final.json")))


def _check(text, limit):
    assert "\n" not in text
    assert len(text) <= limit, len(text)
    line = json.loads(text)
    for k in REQUIRED:
        assert k in line, k
    assert "workload" in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    return line


def test_real_record_fits_and_keeps_the_contract_keys():
    out = _real_record()
    line = _check(bench.compose_headline(out, "bench_detail.json"), bench.LINE_BUDGET)
    assert line["value"] == float("%.5g" % out["value"])
    assert abs(line["roofline"]["frac"] - out["roofline"]["frac"]) < 1e-4 * out["roofline"]["frac"] + 1e-12
    for key in ("pcie_inclusive", "c_abi_streams", "orb_detector", "gp3p", "c3", "c5_epnp", "c5_kneip", "sequence"):
        assert key in line["sub"] and "value" in line["sub"][key], key


def test_worst_case_record_stays_under_the_budget():
    out = _real_record()
    long = "x" * 5000
    out["config"]["workload"] = long
    out["config"]["accuracy_note"] = long
    out["roofline"]["note"] = long
    out["roofline"]["kernel"] = "(" + long + ")"
    out["cpu_baseline"]["sample"] = long
    out["metric"] = out["metric"] + long
    for key in ("orb_detector", "gp3p", "c3", "c5_epnp", "c5_kneip", "sequence", "sequence_rgbd", "pano_1200"):
        out[key] = copy.deepcopy(out.get(key) or {"value": 1.0})
        out[key]["note"] = long
        out[key]["kernels_ms_per_step"] = {"kernel_%d" % i: 0.123456789 for i in range(200)}
        if isinstance(out[key].get("roofline"), dict):
            out[key]["roofline"]["kernel"] = long
    out["c3_2880"] = {"error": long}
    out["sub_error"] = long
    out["kernels_ms_per_step"] = {"kernel_%d" % i: 0.123456789 for i in range(500)}
    text = bench.compose_headline(out, "bench_detail.json")
    _check(text, 6000)
    assert len(text) <= bench.LINE_BUDGET


def test_missing_optional_groups_do_not_break_the_line():
    out = _real_record()
    for key in ("valu_issue", "valu_issue_step", "traffic_step", "cpu_baseline_all_cores", "sequence", "c3", "reference_libraries",
                "accuracy_threshold_0p5deg"):
        out.pop(key, None)
    out["roofline"]["traffic"] = None
    out["value"] = float("nan")
    line = json.loads(bench.compose_headline(out, None))
    assert line["value"] is None and line["roofline"]["traffic"] is None
