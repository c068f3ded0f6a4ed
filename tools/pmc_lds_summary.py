#!/usr/bin/env python3
"""LDS activity and bank conflicts per kernel from a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS` pass
(SURVEY 8d lists SQ_LDS_BANK_CONFLICT among the counters to show):   python tools/pmc_lds_summary.py <counter_collection.csv> [kernel ...]
Values are per launch (mean over the launches in the file), in cycles / instructions summed over the chip."""
import collections
import csv
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, seen = collections.Counter(), set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0].replace("sonic::", "").replace("void ", "").split("<")[0]      # (template arguments dropped: k_part_scatter_staged<false> counts as k_part_scatter_staged)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
    want = sys.argv[2:]
    rows = [(k, v) for k, v in acc.items() if (k in want if want else v.get("SQ_INSTS_LDS", 0) > 0)]
    for k, v in sorted(rows, key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0) / max(1, cnt[kv[0]])):
        n = cnt[k]
        act, conf = v.get("SQ_LDS_IDX_ACTIVE", 0) / n, v.get("SQ_LDS_BANK_CONFLICT", 0) / n
        print(f"{k:26s} launches {n:4d}  SQ_INSTS_LDS {v.get('SQ_INSTS_LDS', 0) / n:11.0f}  SQ_LDS_IDX_ACTIVE {act:12.0f}  SQ_LDS_BANK_CONFLICT {conf:12.0f}  "
              f"conflict / active {conf / max(1.0, act):.3f}")


if __name__ == "__main__":
    main()
