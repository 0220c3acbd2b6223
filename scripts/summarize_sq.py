#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES ... pass of the
bench command into one per-kernel CSV:  python scripts/summarize_sq.py <dir> > profiles/roundN/<tag>_sq_per_kernel.csv
cyc_per_valu = SIMD-cycles available per VALU wave-instruction (duration * 2.4 GHz * 1024 SIMDs / VALU
instructions): ~4.3-4.6 means the kernel sits at the VALU issue limit (scripts/valu_rate.hip), larger values
mean latency / LDS / memory leave issue slots idle."""
import collections
import csv
import glob
import os
import re
import sys


def label_of(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?\(", name)
    if not m:
        return name.split("(")[0][:60]
    base, targs = m.group(1), m.group(2) or ""
    if base == "median_gray_kernel" and "true" in targs:
        return "unwrap_median_gray_kernel"
    return base


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = label_of(row["Kernel_Name"])
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                acc[k]["_dur_us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    w = csv.writer(sys.stdout)
    w.writerow(["label", "dispatches", "avg_us_under_pmc", "valu_insts", "salu_insts", "lds_insts", "vmem_rd_insts", "waves",
                "cyc_per_valu"])
    rows = []
    for k, v in acc.items():
        if not v["SQ_INSTS_VALU"]:
            continue
        mean = lambda c: sum(v[c]) / len(v[c]) if v[c] else 0.0  # noqa: E731
        dur, valu = mean("_dur_us"), mean("SQ_INSTS_VALU")
        rows.append((k, len(v["SQ_INSTS_VALU"]), dur, valu, mean("SQ_INSTS_SALU"), mean("SQ_INSTS_LDS"), mean("SQ_INSTS_VMEM_RD"),
                     mean("SQ_WAVES"), dur * 1e-6 * 2.4e9 * 1024 / valu if valu else 0.0))
    for r in sorted(rows, key=lambda r: -r[2] * r[1]):
        w.writerow([r[0], r[1], "%.1f" % r[2], "%.0f" % r[3], "%.0f" % r[4], "%.0f" % r[5], "%.0f" % r[6], "%.0f" % r[7], "%.2f" % r[8]])


if __name__ == "__main__":
    main()
