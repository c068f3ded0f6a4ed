#!/usr/bin/env python3
"""Static instruction count of one bucket-walk iteration of k_bucket_accum (one XYZZ mixed addition), from the gfx950 ISA
hipcc emits: the hot basic blocks of the loop (the ones on the path through the block with ten Montgomery-routine calls,
excluding the doubling / infinity side paths) plus 10 x the routine's length.  bench.py uses the result as the VALU-issue
model of the kernel (every VALU instruction of a wave64 occupies its SIMD for 4 cycles; one wave per SIMD already issues
back to back, so instructions, not latency, are what the kernel pays for).

    python tools/count_accum_instrs.py          # compiles sonic_amd/csrc/msm.hip to ISA in /tmp and prints the counts"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
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
    # the hot block is the fused mixed addition: the one that calls the two-product core (else: the block with ten product calls)
    # (nine calls: six products, two squarings, one two-product; the affine + affine statement before the loop has five)
    fused = [k for k, b in enumerate(blocks) if any("sonic_mont_mul2_fq_core@rel32" in x for x in b[1]) and calls[k] == 9]
    hot = fused[0] if fused else calls.index(10)
    # innermost loop around the hot block: smallest [target, source] span of a backward branch that contains it
    span = None
    for k, b in enumerate(blocks):
        for x in b[1]:
            if x.startswith("s_cbranch") or x.startswith("s_branch"):
                t = index.get(x.split()[-1])
                if t is not None and t <= hot <= k and (span is None or k - t < span[1] - span[0]):
                    span = (t, k)
    body = [b for k, b in enumerate(blocks) if span[0] <= k <= span[1] and (k == hot or calls[k] == 0)]   # drops the doubling side path
    outside = sum(len(b[1]) for b in body)
    # the routines the hot block calls, by symbol, and their lengths
    targets = collections.Counter(m.group(1) for x in blocks[hot][1] for m in [re.search(r"(sonic_mont_\w+)@rel32@lo", x)] if m)

    def routine_len(callee):
        i0 = next(i for i, l in enumerate(L) if l.strip() == callee + ":")
        i1 = next(i for i in range(i0, len(L)) if L[i].strip().startswith(".size\t" + callee) or L[i].strip().startswith(".size " + callee))
        body_ = [l.strip() for l in L[i0 + 1:i1] if l.strip() and not l.strip().startswith((";", "."))]
        return len(body_), sum(1 for l in body_ if l.startswith("v_mad_u64_u32"))
    total, mads = outside, sum(1 for b in body for x in b[1] if x.startswith("v_mad_u64_u32"))
    print("blocks on the hot path:", [(b[0], len(b[1])) for b in body])
    print("instructions outside the Montgomery routines per mixed addition:", outside)
    for callee, cnt in sorted(targets.items()):
        n, m = routine_len(callee)
        print(f"routine: {callee} {n} instructions ({m} v_mad_u64_u32) x {cnt}")
        total += n * cnt
        mads += m * cnt
    print("total per mixed addition:", total, "of which v_mad_u64_u32:", mads)


if __name__ == "__main__":
    sys.exit(main())
