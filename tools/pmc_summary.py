#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes of `bench.py --msm-only` (FETCH_SIZE, WRITE_SIZE; one counter per pass, as
MI355X_MICROARCH.md's HBM section prescribes) into profiles/<round>_pmc_msm.json: HBM bytes per launch of every MSM kernel.

    python tools/pmc_summary.py <FETCH_SIZE_counter_collection.csv> <WRITE_SIZE_counter_collection.csv> <msm_n> <c> <W> <sets> > out.json

Units and the gfx950 correction (same guide): both counters are in KB; FETCH_SIZE counts a 128-B request as 64 B on
gfx950, so reads are doubled.  The last 3 launches of each kernel are the bench's timed MSMs (warm-up comes first)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("sonic::", "").replace("void ", "").split("<")[0]
        vals[name].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: sorted(v) for k, v in vals.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    msm_n, c, W, sets = (int(x) for x in sys.argv[3:7])
    out = {"command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --output-format csv -- python3 bench.py --msm-only --msm-lanes 0 --no-cpu --steps 3 "
                      "--warmup 1  (one counter per pass; values = mean over the launches of the 3 timed MSMs)",
           "msm_n": msm_n, "plan": {"window_bits": c, "windows": W, "bucket_sets": sets},
           "method": "MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 128-B requests "
                     "as 64 B, so reads are doubled"}
    msm_kernels = ["k_part_hist", "k_part_scatter", "k_part_scatter_staged", "k_part_sort", "k_border_hist", "k_border_scatter", "k_border_place", "k_bucket_accum",
                   "k_heavy_accum", "k_heavy_finish", "k_bucket_segments", "k_group_sum", "k_window_sum", "k_bucket_tree_block", "k_bucket_tree_levels"]
    total = 0
    for k in msm_kernels:
        if k not in fetch or k not in write:
            continue
        per_msm = max(1, len(fetch[k]) // 4)                     # launches per MSM (1 warm-up + 3 timed)
        f = [v for _, v in fetch[k][-3 * per_msm:]]
        w = [v for _, v in write[k][-3 * per_msm:]]
        fkb, wkb = sum(f) / 3.0, sum(w) / 3.0                     # per MSM
        b = int(round((2.0 * fkb + wkb) * 1024))
        total += b
        out[k] = {"FETCH_SIZE_KB": round(fkb), "WRITE_SIZE_KB": round(wkb), "hbm_bytes_per_launch": b, "launches_per_msm": per_msm}
    out["all_msm_kernels_hbm_bytes_per_msm"] = total
    out["algorithmic_bytes_per_msm"] = {"survey_8d": 128 * msm_n,
                                        "with_window_tables": (32 + 96 * W) * msm_n,
                                        "note": "the table method reads one 96-B table point per (term, window): 32 + 96 W bytes per term"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
