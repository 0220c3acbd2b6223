"""Prints the headline numbers of a bench.py record: the FULL record (bench_detail.json, which holds the per-kernel table)
or a captured stdout whose last line is the compact line: value, ms per step, per-kernel ms."""
import json
import sys

text = open(sys.argv[1]).read().strip()
try:
    d = json.loads(text)
except ValueError:
    d = json.loads(text.splitlines()[-1])
print("%.0f %s  %.3f ms/step  streams=%s B=%s" % (d["value"], d["unit"], d["ms_per_step"], d.get("streams"),
                                                   d["config"].get("pairs_per_gpu")))
for k, v in sorted(d.get("kernels_ms_per_step", {}).items(), key=lambda kv: -kv[1]):
    print("  %-42s %.3f" % (k, v))
print("  total %.3f" % d.get("kernel_ms_per_step_total", 0.0))
for k in ("roofline", "valu_issue", "cpu_baseline", "pcie_inclusive"):
    if k in d and d[k]:
        print(k, {a: b for a, b in d[k].items() if a not in ("note", "sample", "traffic_source", "source")})
