#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv / counter_collection.csv into a short table (names truncated)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("craft::", "")
    return name[:60]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    if "Counter_Name" in rows[0]:
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            print(k)
            for c, v in d.items():
                print(f"    {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
    else:
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print(f"total kernel time {tot / 1e6:.2f} ms")
        for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
            print(f"{short(r['Name']):60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f} %")


if __name__ == "__main__":
    main()
