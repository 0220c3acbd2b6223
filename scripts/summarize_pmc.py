#!/usr/bin/env python3
"""Condense two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) of the
bench command into one per-kernel CSV for profiles/:

    python scripts/summarize_pmc.py <fetch_dir> <write_dir> <pairs_per_launch> > profiles/roundN/..._pmc_hbm_per_kernel.csv

Columns: label (the library's profile label, what bench.py's roofline names), dispatches, raw counter averages in
KB per dispatch, and hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section:
on gfx950 FETCH_SIZE tallies 64 B per 128-B request; WRITE_SIZE is exact)."""
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


def collect(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                a = acc.setdefault(label_of(row["Kernel_Name"]), [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    w = csv.writer(sys.stdout)
    w.writerow(["label", "dispatches_fetch_pass", "FETCH_SIZE_KB_avg_raw", "dispatches_write_pass", "WRITE_SIZE_KB_avg_raw",
                "hbm_bytes_per_launch", "pairs_per_launch"])
    rows = []
    for k in sorted(set(fetch) | set(write)):
        nf, sf = fetch.get(k, [0, 0.0])
        nw, sw = write.get(k, [0, 0.0])
        f, wr = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        rows.append((k, nf, f, nw, wr, (2.0 * f + wr) * 1024.0))
    for r in sorted(rows, key=lambda r: -r[5]):
        w.writerow([r[0], r[1], "%.3f" % r[2], r[3], "%.3f" % r[4], "%.0f" % r[5], pairs])


if __name__ == "__main__":
    main()
