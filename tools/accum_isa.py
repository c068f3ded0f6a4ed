#!/usr/bin/env python3
"""Per-category instruction budget of ONE bucket-walk addition of k_bucket_accum (the hot loop body + the Montgomery routines it calls),
from the gfx950 ISA hipcc emits, and what each category costs at the issue rates tools/microbench measured on the chip
(profiles/rNN_microbench.txt).  The point: show where the kernel's cycles go, and that the register-only loop of the same addition
already runs at the rate the instruction mix allows (so nothing that only hides LATENCY -- more waves, two buckets per thread
interleaved -- can make it faster; only fewer or cheaper instructions can).

    python tools/accum_isa.py [profiles/r05_microbench.txt] > profiles/r06_accum_isa.txt"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CATS = [
    ("v_mad_u64_u32 (32x32+64 multiply-add)", lambda o: o.startswith("v_mad_u64_u32")),
    ("carry arithmetic (v_add_co / v_addc_co / v_sub_co / v_subb_co / v_add_u32 / v_sub_u32 / v_add3 / v_lshl_add ...)", lambda o: re.match(r"v_(add|addc|sub|subb|subrev|subbrev|add3|lshl_add|add_lshl)", o) is not None),
    ("32-bit multiplies (v_mul_lo_u32 / v_mul_hi_u32: the Montgomery quotient digits)", lambda o: o.startswith("v_mul_")),
    ("moves and selects (v_mov / v_cndmask / v_accvgpr / v_readlane / v_readfirstlane / v_perm / v_swap)", lambda o: re.match(r"v_(mov|cndmask|accvgpr|readlane|readfirstlane|writelane|perm|swap|bfi|bfe|and|or|xor|not|lshl|lshr|ashr|alignbit)", o) is not None),
    ("vector compares (v_cmp*)", lambda o: o.startswith("v_cmp")),
    ("memory (global_load / global_store / flat / buffer / ds / scratch)", lambda o: re.match(r"(global_|flat_|buffer_|ds_|scratch_)", o) is not None),
    ("scalar ALU and branches (s_* except waits)", lambda o: o.startswith("s_") and not re.match(r"s_(nop|waitcnt|sleep)", o)),
    ("s_nop / s_waitcnt", lambda o: re.match(r"s_(nop|waitcnt)", o) is not None),
]


def categorise(lines):
    c = collections.Counter()
    for l in lines:
        op = l.split()[0]
        for name, pred in CATS:
            if pred(op):
                c[name] += 1
                break
        else:
            c["other: " + op] += 1
    return c


def main():
    mb = sys.argv[1] if len(sys.argv) > 1 else sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_microbench.txt"))[-1]
    mb = mb if os.path.isabs(mb) or os.path.exists(mb) else os.path.join(ROOT, "profiles", mb)
    src = os.path.join(ROOT, "sonic_amd", "csrc", "msm.hip")
    out = "/tmp/sonic_msm_isa.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(src))
    L = open(out).read().split("\n")
    start = next(i for i, l in enumerate(L) if l.startswith("_ZN5sonic14k_bucket_accum") and ":" in l)
    end = next(i for i in range(start, len(L)) if L[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", []]
    for i in range(start + 1, end + 1):
        l = L[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        elif l and not l.startswith(";") and not l.startswith("."):
            cur[1].append(l)
    blocks.append(cur)
    calls = [sum(1 for x in b[1] if x.startswith("s_swappc")) for b in blocks]
    index = {b[0]: k for k, b in enumerate(blocks)}
    fused = [k for k, b in enumerate(blocks) if any("sonic_mont_mul2_fq_core@rel32" in x for x in b[1]) and calls[k] == 9]
    hot = fused[0] if fused else calls.index(10)
    span = None
    for k, b in enumerate(blocks):
        for x in b[1]:
            if x.startswith("s_cbranch") or x.startswith("s_branch"):
                t = index.get(x.split()[-1])
                if t is not None and t <= hot <= k and (span is None or k - t < span[1] - span[0]):
                    span = (t, k)
    body = [b for k, b in enumerate(blocks) if span[0] <= k <= span[1] and (k == hot or calls[k] == 0)]
    outside = [x for b in body for x in b[1]]
    targets = collections.Counter(m.group(1) for x in blocks[hot][1] for m in [re.search(r"(sonic_mont_\w+)@rel32@lo", x)] if m)

    def routine(callee):
        i0 = next(i for i, l in enumerate(L) if l.strip() == callee + ":")
        i1 = next(i for i in range(i0, len(L)) if L[i].strip().startswith(".size\t" + callee) or L[i].strip().startswith(".size " + callee))
        return [l.strip() for l in L[i0 + 1:i1] if l.strip() and not l.strip().startswith((";", "."))]

    total = categorise(outside)
    print("# tools/accum_isa.py: one bucket-walk addition of k_bucket_accum (XYZZ + affine, madd-2008-s: 6 products, 2 squarings, 1 two-product call),")
    print("# static count of the gfx950 ISA of the committed sonic_amd/csrc/msm.hip; rates from", os.path.relpath(mb, ROOT))
    print()
    print("loop body outside the Montgomery routines (field additions / subtractions / negation of q.y, the gather of the next point, loop control): %d instructions" % len(outside))
    for name, cnt in sorted(categorise(outside).items(), key=lambda kv: -kv[1]):
        print("    %5d  %s" % (cnt, name))
    for callee, cnt in sorted(targets.items()):
        body_ = routine(callee)
        c = categorise(body_)
        print("routine %s: %d instructions, called %d x per addition" % (callee, len(body_), cnt))
        for name, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print("    %5d  %s" % (v, name))
        for k_, v in c.items():
            total[k_] += v * cnt
    n_all = sum(total.values())
    print()
    print("per addition, all of it: %d instructions" % n_all)
    # issue cost per wave64 instruction on one SIMD, from the microbenchmarks (lane-ops per clock per CU / 4 SIMDs)
    txt = open(mb).read()
    mad = float(re.search(r"v_mad_u64_u32:.*\(([\d.]+) lane-ops/clk/CU", txt).group(1))
    simple = 3 * float(re.search(r"add\+xor\+shift:.*\(([\d.]+) lane-triples/clk/CU", txt).group(1))
    mul32 = float(re.search(r"mul_lo\+add:.*\(([\d.]+) lane-ops/clk/CU", txt).group(1))
    walk = float(re.search(r"g1_add_mixed_walk \(fused asm\):\s+[\d.]+ ms -> ([\d.e+]+) add/s", txt).group(1))
    cyc = lambda rate: 64.0 / (rate / 4.0)          # cycles a wave64 instruction of that class occupies its SIMD  # noqa: E731
    cost = {}
    for name, cnt in total.items():
        if name.startswith("v_mad_u64_u32"):
            per = cyc(mad)
        elif name.startswith("32-bit multiplies"):
            per = cyc(2 * mul32)      # (the microbenchmark's pair is a multiply and an add)
        elif name.startswith(("memory", "scalar", "s_nop")):
            per = 0.0                  # issued by other ports / overlapped; counted as free here
        else:
            per = cyc(simple)
        cost[name] = cnt * per
    cyc_total = sum(cost.values())
    print("    %-112s %6s %8s %7s" % ("category", "count", "cycles", "share"))
    for name, cnt in sorted(total.items(), key=lambda kv: -cost[kv[0]]):
        print("    %-112s %6d %8.0f %6.1f%%" % (name, cnt, cost[name], 100 * cost[name] / cyc_total))
    print("    issue cycles per wave-addition on one SIMD: %.0f (%.2f us at 2.4 GHz); the chip has 1024 SIMDs x 64 lanes:" % (cyc_total, cyc_total / 2.4e3))
    bound = 1024 * 64 / (cyc_total / 2.4e9)
    print("    issue-bound rate of this instruction mix: %.3e additions/s" % bound)
    print("    measured, the same fused addition in a register-only loop at 2 waves per SIMD (tools/microbench): %.3e additions/s = %.1f %% of that bound" % (walk, 100 * walk / bound))
    madc = [v for k_, v in total.items() if k_.startswith("v_mad_u64_u32")][0]
    print("    v_mad_u64_u32 alone: %d per addition = %.1f %% of the issue cycles: the MAD roof (int_roofline) can be reached to that fraction at most" % (madc, 100 * [cost[k_] for k_ in cost if k_.startswith("v_mad_u64_u32")][0] / cyc_total))


if __name__ == "__main__":
    main()
