#!/usr/bin/env python3
"""Writes profiles/<round>_kernel_model.json: the numbers bench.py's int_roofline is computed from, each with its source --
  * instructions and v_mad_u64_u32 per bucket-walk addition: static count of the ISA hipcc emits for k_bucket_accum
    (tools/count_accum_instrs.py, run here; no GPU needed);
  * the chip's v_mad_u64_u32 issue rate and the rate of the same fused addition in a register-only loop: parsed from the
    on-hardware microbenchmark output (tools/microbench > profiles/<round>_microbench.txt).
bench.py reads the newest profiles/rNN_kernel_model.json; nothing of this is a constant in bench.py.

    python tools/kernel_model.py r03 [profiles/r03_microbench.txt]"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    mb = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", f"{tag}_microbench.txt")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "count_accum_instrs.py")], capture_output=True, text=True, check=True).stdout
    m = re.search(r"total per mixed addition: (\d+) of which v_mad_u64_u32: (\d+)", out)
    if not m:
        sys.exit("count_accum_instrs.py output not understood:\n" + out)
    instr, mads = int(m.group(1)), int(m.group(2))
    txt = open(mb).read()
    mad = re.search(r"v_mad_u64_u32:\s+[\d.]+ ms -> ([\d.e+]+) mad/s", txt)
    walk = re.search(r"g1_add_mixed_walk \(fused asm\):\s+[\d.]+ ms -> ([\d.e+]+) add/s", txt)
    fq = re.search(r"fq_mul:\s+[\d.]+ ms -> ([\d.e+]+) mul/s", txt)
    if not (mad and walk):
        sys.exit(f"{mb}: microbenchmark lines not found")
    model = {
        "round": tag,
        "instr_per_addition": instr, "mads_per_addition": mads,
        "instr_source": "tools/count_accum_instrs.py on the committed sonic_amd/csrc/msm.hip (static ISA count of one bucket-walk iteration: the hot "
                        "blocks of the loop + the Montgomery routines it calls)",
        "instr_breakdown": [l for l in out.splitlines() if l.startswith("routine:") or l.startswith("instructions outside")],
        "mad_peak_per_s": float(mad.group(1)), "addition_register_loop_per_s": float(walk.group(1)),
        "fq_mul_per_s": float(fq.group(1)) if fq else None,
        "rate_source": os.path.relpath(mb, ROOT) + " (tools/microbench on one MI355X: independent v_mad_u64_u32 chains on all CUs; g1_add_mixed_walk = the "
                       "kernel's fused addition in a loop without memory accesses)",
    }
    path = os.path.join(ROOT, "profiles", f"{tag}_kernel_model.json")
    json.dump(model, open(path, "w"), indent=1)
    print(json.dumps(model, indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
