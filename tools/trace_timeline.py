#!/usr/bin/env python3
"""Timeline of the LAST forward pass in a rocprofv3 kernel trace (…_kernel_trace.csv): every kernel with its queue, start offset and
duration, plus the phase sums (encoders / transformer + correlation / refinement loop) and one iteration of the loop.
    python tools/trace_timeline.py <kernel_trace.csv> [first_kernel_substring=k_stem_mfma]"""
import csv, re, sys


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("craft::", "")[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows))
    # the last forward: from the last-but-one k_stem_mfma pair backwards -- find the starts of forwards = a stem after >= 200 other kernels
    stems = [i for i, e in enumerate(ev) if "k_stem_mfma" in e[3]]
    starts = [stems[0]]
    for i in stems[1:]:
        if i - starts[-1] > 200:
            starts.append(i)
    starts.append(len(ev))
    # the forward with the most kernels (a full 12-iteration pass; the roofline legs of bench.py run shorter ones), the latest of those
    spans = [(starts[k + 1] - starts[k], k) for k in range(len(starts) - 1)]
    best = max(n for n, _ in spans)
    k = max(k for n, k in spans if n >= best - 8)
    fw = ev[starts[k]:starts[k + 1]]
    t0 = fw[0][0]
    qs = sorted({e[2] for e in fw})
    print(f"# forward: {len(fw)} kernels, {(fw[-1][1] - t0) / 1e3:.1f} us wall, queues {qs}")
    for s, e, q, n in fw:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{qs.index(q)}  {n}")


if __name__ == "__main__":
    main()
