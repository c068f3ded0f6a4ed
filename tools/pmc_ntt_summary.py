#!/usr/bin/env python3
"""HBM bytes per launch of the NTT kernels from the two rocprofv3 --pmc passes of tools/ntt_time.py (FETCH_SIZE, WRITE_SIZE; one
counter per pass, MI355X_MICROARCH.md's HBM section) and their durations from the --kernel-trace --stats pass:
    python tools/pmc_ntt_summary.py <FETCH csv> <WRITE csv> <kernel_stats csv> > profiles/rNN_pmc_ntt.json
Units and the gfx950 correction as in tools/pmc_summary.py: both counters are in KB, FETCH_SIZE counts a 128-B request as 64 B."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            vals[r["Kernel_Name"].split("(")[0].replace("sonic::", "")].append(float(r["Counter_Value"]))
    return vals


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    stats = {r["Name"].split("(")[0].replace("sonic::", ""): r for r in csv.DictReader(open(sys.argv[3]))}
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE; then --kernel-trace --stats) -- python3 tools/ntt_time.py  (13 products of M = 2^21: "
                      "3 warm-up + 10 timed; values = mean over all launches)",
           "method": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so reads are doubled",
           "M": 1 << 21}
    total = 0.0
    # round 5: the ten wide stages of a 2^21-point transform run as ONE pass (k_ntt_wide_big): 3 + 3 launches per product instead of 6 + 3
    launches = {"k_ntt_wide_big": 3, "k_ntt_wide": 6, "k_ntt_local": 3, "k_ntt_wide4": 6, "k_ntt_local4": 3}
    for k, per_product in launches.items():
        if k not in fetch or k not in write:
            continue
        f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
        b = (2.0 * f + w) * 1024
        out[k] = {"FETCH_SIZE_KB": round(f), "WRITE_SIZE_KB": round(w), "hbm_bytes_per_launch": int(round(b)), "launches_per_product": per_product,
                  "rocprof_avg_us": round(float(stats[k]["AverageNs"]) / 1e3, 2) if k in stats else None}
        total += b * per_product
    out["hbm_bytes_per_product"] = int(round(total))
    passes = 2 if "k_ntt_wide_big" in out else 3
    out["hbm_passes_per_transform"] = passes
    out["bytes_by_design_per_product"] = (3 * passes * 64 + 96) * (1 << 21)
    out["algorithmic_bytes_per_product"] = 288 * (1 << 21)
    used = [k for k in launches if out.get(k, {}).get("rocprof_avg_us")]
    if used:
        out["rocprof_ms_per_product"] = round(sum(launches[k] * out[k]["rocprof_avg_us"] for k in used) / 1e3, 4)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
