#!/usr/bin/env python3
"""Per-queue listing of one prove() (the last but one) in a rocprofv3 --kernel-trace CSV: start, duration and gap to the previous kernel
of the same queue.  usage: python tools/perqueue.py trace.csv [min_us] [which: -2]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sonic::", "").replace("void ", "")[:26], r["Queue_Id"],
                 r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))) for r in rows)
    starts = [k[0] for k in ks if "k_build_r1" in k[2]]
    s0 = starts[which]
    s1 = starts[which + 1] if which + 1 < 0 and len(starts) > 1 else None
    sel = [k for k in ks if k[0] >= s0 and (s1 is None or k[0] < s1)]
    if s1 is None:      # a single proof: until the device pauses for 3 ms
        out, last = [], s0
        for k in sel:
            if out and k[0] - last > 3_000_000:
                break
            out.append(k)
            last = max(last, k[1])
        sel = out
    print("span %.3f ms, %d kernels, sum of durations %.3f ms" % ((max(k[1] for k in sel) - s0) / 1e6, len(sel), sum(k[1] - k[0] for k in sel) / 1e6))
    byq = collections.defaultdict(list)
    for k in sel:
        byq[k[3]].append(k)
    for q, l in sorted(byq.items()):
        print("== queue", q, len(l))
        prev = None
        for s, e, n, _, g, w in l:
            gap = (s - prev) / 1e3 if prev else 0
            if (e - s) / 1e3 >= min_us or gap > 20:
                print("  %7.3f  dur %7.1f us  gap %6.1f  %-26s grid %s wg %s" % ((s - s0) / 1e6, (e - s) / 1e3, gap, n, g, w))
            prev = e


if __name__ == "__main__":
    main()
