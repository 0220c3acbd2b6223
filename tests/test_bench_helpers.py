"""CPU: the pieces of bench.py that turn the committed rocprofv3 summaries (profiles/) into the roofline fields of the
JSON line -- they must keep reading the files the profile scripts write."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profile_summaries_feed_the_roofline_fields():
    b = _bench()
    traffic, src = b.pmc_traffic("unwrap_median_gray_kernel", 256)
    assert src and src.startswith("profiles/") and 4e8 < traffic < 1.5e9          # ~700 MB per launch of 256 pairs
    t2, _ = b.pmc_traffic("unwrap_median_gray_kernel", 128)
    assert abs(t2 * 2 - traffic) < 1e-6 * traffic                                 # scales with the pairs per launch
    assert b.pmc_traffic("no_such_kernel", 256) == (None, None)
    vi = b.valu_issue("unwrap_median_gray_kernel", 256, 2.42e-3)       # round-2 launch time
    assert vi["mix_source"] and 3.0 < vi["issue_cycles_per_inst"] < 4.4 and 0.8 < vi["frac"] < 1.05
    step = b.valu_issue_step(768, 14.1)
    assert 0.55 < step["dominant_kernel_share"] < 0.75 and 0.7 < step["frac"] < 1.05
    assert 0.4 < step["frac_vs_guide_2cyc"] < step["frac"] and step["per_kernel"] and step["mix_source"]   # priced per kernel
    meta = b.profile_meta(os.path.join(ROOT, step["source"]), {"detector": "GFT"})
    assert "stale" in meta
    assert b.b_alg_c2(480, 640, 2000) == 1943296                                   # algorithmic bytes per pair
    assert b.host_cores() >= 1
