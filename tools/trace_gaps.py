#!/usr/bin/env python3
"""Idle-time analysis of a rocprofv3 kernel trace (…_kernel_trace.csv): per queue and for the union of all queues, how much of the
wall time between the first and the last kernel no kernel was running, and which kernel boundaries carry the largest gaps.
    python tools/trace_gaps.py <kernel_trace.csv> [last_ms]       (last_ms: analyse only the final window, e.g. the timed steps)"""
import csv, re, sys
from collections import defaultdict


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("craft::", "")[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows))
    if len(sys.argv) > 2:
        t0 = ev[-1][1] - float(sys.argv[2]) * 1e6
        ev = [e for e in ev if e[0] >= t0]
    wall = ev[-1][1] - ev[0][0]
    # union busy
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    gaps = defaultdict(lambda: [0, 0])
    prev_name = ev[0][3]
    for s, e, q, n in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps[(prev_name, n)][0] += s - cur_e
            gaps[(prev_name, n)][1] += 1
            cur_s, cur_e = s, e
            prev_name = n
        else:
            if e > cur_e:
                cur_e = e
                prev_name = n
    busy += cur_e - cur_s
    print(f"kernels {len(ev)}  wall {wall / 1e6:.3f} ms  union busy {busy / 1e6:.3f} ms  idle {(wall - busy) / 1e6:.3f} ms ({100.0 * (wall - busy) / wall:.1f} %)")
    perq = defaultdict(int)
    for s, e, q, n in ev:
        perq[q] += e - s
    for q, b in sorted(perq.items(), key=lambda x: -x[1]):
        print(f"  queue {q}: busy {b / 1e6:.3f} ms")
    print("largest idle boundaries (after kernel -> before kernel): total us, count, avg us")
    for (a, b), (t, c) in sorted(gaps.items(), key=lambda x: -x[1][0])[:25]:
        print(f"  {a:40s} -> {b:40s} {t / 1e3:9.1f} {c:5d} {t / 1e3 / c:7.2f}")


def overlap(path, pattern, last_ms=None):
    """For every kernel whose name matches ``pattern``: how much of its duration ran beside kernels of OTHER queues (side-stream overlap)."""
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows))
    if last_ms:
        t0 = ev[-1][1] - float(last_ms) * 1e6
        ev = [e for e in ev if e[0] >= t0]
    tot = ov = n = 0
    beside = defaultdict(int)
    for s, e, q, name in ev:
        if not re.search(pattern, name):
            continue
        n += 1
        tot += e - s
        segs = sorted((max(s, s2), min(e, e2), n2) for s2, e2, q2, n2 in ev if q2 != q and s2 < e and e2 > s)
        cur = s
        for a, b, n2 in segs:                     # union of the other queues' kernels inside [s, e]
            if b > cur:
                ov += b - max(a, cur)
                beside[n2] += b - max(a, cur)
                cur = b
    print(f"{pattern}: {n} launches, {tot / 1e3 / max(n, 1):.1f} us each; {100.0 * ov / max(tot, 1):.1f} % of their time ran beside a kernel of another queue")
    for k, v in sorted(beside.items(), key=lambda kv: -kv[1])[:6]:
        print(f"     beside {k:44s} {100.0 * v / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--overlap":      # trace_gaps.py <csv> --overlap <regex> [last_ms]
        overlap(sys.argv[1], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        main()
