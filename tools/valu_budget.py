#!/usr/bin/env python3
"""VALU instruction budget of prove() per kernel, from a rocprofv3 --pmc SQ_INSTS_VALU pass:
    rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d out -o t -- python3 bench.py --no-cpu --no-pipeline --prove-only --steps 2 --warmup 1
    python tools/valu_budget.py out/t_counter_collection.csv <proofs in the run: warm-up + timed + 1 (the sequential leg runs none with --no-pipeline)>
Streamed proofs are bound by instruction issue, so this -- not kernel durations, which overlap -- is where a proof's time goes."""
import collections
import csv
import sys


def main():
    tot = collections.Counter()
    cnt = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        name = r["Kernel_Name"].split("(")[0].replace("sonic::", "").replace("void ", "").split("<")[0]
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    for setup in ("k_table_step", "k_srs_points", "k_batch_affine", "k_fb_table", "k_xyzz_to_affine", "k_ntt_twiddles", "k_fr_inv_pow2",
                  "k_fr_setup_x", "k_weight_row_poly"):      # SRS.new / handle set-up, not part of a proof
        tot.pop(setup, None)
        cnt.pop(setup, None)
    proofs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    allv = sum(tot.values())
    print(f"# wave-level VALU instructions per proof (total {allv / proofs:.3e}), share, launches per proof; the run's few set-up MSMs (prepare, the MSM leg) are spread over the proofs")
    for name, v in tot.most_common():
        print(f"{name:28s} {v / proofs:12.4e} {100 * v / allv:6.2f} %  {cnt[name] / proofs:7.1f}")


if __name__ == "__main__":
    main()
