#!/usr/bin/env python3
"""Per-kernel HBM-side traffic table from three rocprofv3 passes over the same command:
    python tools/traffic_table.py <kernel_stats.csv> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv>
FETCH_SIZE / WRITE_SIZE are in KiB at the L2 <-> fabric boundary (MALL hits included); on gfx950 FETCH_SIZE counts 64-byte
requests as 32 (MI355X_MICROARCH.md, rocprofv3 section), hence the factor 2."""
import csv, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("craft::", "")[:56]


def counters(path):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    stats = list(csv.DictReader(open(sys.argv[1])))
    f, w = counters(sys.argv[2]), counters(sys.argv[3])
    print(f"{'kernel':56s} {'calls':>6s} {'avg us':>8s} {'fetch MB':>9s} {'write MB':>9s} {'TB/s':>6s}")
    for r in stats[: int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
        k = short(r["Name"])
        us = float(r["AverageNs"]) / 1e3
        fb, wb = f.get(k, 0.0) * 1024 * 2, w.get(k, 0.0) * 1024
        print(f"{k:56s} {int(r['Calls']):6d} {us:8.1f} {fb / 1e6:9.1f} {wb / 1e6:9.1f} {(fb + wb) / us / 1e6:6.2f}")


if __name__ == "__main__":
    main()
