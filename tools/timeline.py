#!/usr/bin/env python3
"""Timeline of one prove() (the last but one) in a rocprofv3 --kernel-trace CSV: device span, union busy time, concurrency histogram
and the long kernels in start order.  usage: python tools/timeline.py gpurun_out/trace/*/*kernel_trace.csv [min_us]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_ns = int(float(sys.argv[2]) * 1e3) if len(sys.argv) > 2 else 300_000
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("sonic::", "")[:28],
                 r.get("Queue_Id", "?")) for r in rows)
    starts = [s for s, e, n, q in ks if "k_build_r1" in n]
    # the last-but-one proof: everything from its first kernel to the first kernel of the next proof (run the bench with
    # --no-pipeline so that consecutive proofs do not overlap); with a single proof: until the device pauses for 2 ms
    s0, s1 = (starts[-2], starts[-1]) if len(starts) >= 2 else (starts[-1], None)
    sel, last_end = [], s0
    for s, e, n, q in (k for k in ks if k[0] >= s0):
        if (s1 is not None and s >= s1) or (s1 is None and sel and s - last_end > 2_000_000):
            break
        sel.append((s, e, n, q))
        last_end = max(last_end, e)
    print("device span %.2f ms, %d kernels" % ((last_end - s0) / 1e6, len(sel)))
    ev = sorted([(s, 1) for s, e, n, q in sel] + [(e, -1) for s, e, n, q in sel])
    busy, cur, prev, conc = 0, 0, ev[0][0], collections.Counter()
    for t, d in ev:
        if cur > 0:
            busy += t - prev
        conc[cur] += t - prev
        cur += d
        prev = t
    print("busy %.2f ms; ms at concurrency k: %s" % (busy / 1e6, {k: round(v / 1e6, 2) for k, v in sorted(conc.items())}))
    for s, e, n, q in sel:
        if e - s > min_ns:
            print("%8.2f %8.2f  q%-3s %s" % ((s - s0) / 1e6, (e - s) / 1e6, q, n))


if __name__ == "__main__":
    main()
