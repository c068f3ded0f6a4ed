"""CPU model of the generated gfx950 assembly (tools/gen_mont_asm.py -> sonic_amd/csrc/mont_asm.hpp).

The Montgomery products, the add / sub blocks and the fused mixed addition are straight-line programs over a small
instruction subset, so they can be executed here on python integers, one lane at a time, without a GPU:
  * the product routines give a * b * R^-1 (mod p): canonical for Fr, in the lazy range [0, 2q) for Fq, also for
    operands taken from [q, 2q); the "core" variant gives the same when its caller provides the zero halves;
  * no VALU instruction reads a carry (VCC / SGPR pair) within two issue slots of the VALU write that produced it
    (the gfx950 hazard the schedules are built around);
  * the fused mixed addition equals the XYZZ formulas (madd-2008-s) on random operands in both representatives, leaves
    the accumulator untouched and raises its flag for every exceptional lane (infinity operand, P = +-Q).
What the real hardware does with these instructions is checked by tools/microbench.hip and the -m gpu tests."""
import os
import random
import re
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import gen_mont_asm as G  # noqa: E402

M32 = 0xFFFFFFFF


class Lane:
    """One lane of a wave: 32-bit VGPRs / SGPRs, 1-bit carries, EXEC as a boolean."""

    def __init__(self, core_prog=None):
        self.v, self.s, self.c, self.exec, self.saved = {}, {}, {}, True, {}
        self.core_prog = core_prog      # one routine, or {symbol: routine}
        self.target, self.calls = None, []
        self.trace = []          # (kind, carry-register) per issued instruction, for the hazard check
        self.lds, self.glob, self.s64 = {}, {}, {}      # byte address -> 32-bit word (the NTT butterflies' operands); 64-bit SGPR pairs
        self.loads_in_flight = []         # registers a load has been issued for and no s_waitcnt has covered yet
        self.store_reads = []             # (slot, registers) of LDS stores, for the data-hold check
        self.reg_writes = []              # (slot, register)

    def rd(self, tok):
        tok = tok.strip()
        if tok.startswith("0x"):
            return int(tok, 16)
        if tok.isdigit():
            return int(tok)
        if tok[0] == "s":
            return self.s[tok]
        val = self.v.get(tok, 0xDEADBEEF)
        assert not isinstance(val, tuple), f"{tok} is read before the s_waitcnt that covers its load"
        return val

    def wr(self, tok, val):
        if self.exec:
            self.v[tok.strip()] = val & M32
            self.reg_writes.append((len(self.trace), tok.strip()))

    def pair(self, tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok.strip())
        return f"v{m.group(1)}", f"v{m.group(2)}"

    def run(self, prog):
        for ins in prog:
            self.step(ins)

    def step(self, ins):
        op, _, rest = ins.partition(" ")
        a = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
        carry_read, carry_write = None, None
        if op == "s_nop":
            self.trace.extend([("nop", None, None)] * (int(a[0]) + 1))
            return
        if op == "s_add_u32":                        # the call sequence names its target routine here
            m = re.search(r"(sonic_mont_\w+)@rel32", rest)
            if m:
                self.target = m.group(1)
            return
        if op in ("s_setpc_b64", "s_getpc_b64", "s_addc_u32"):
            return
        if op == "s_swappc_b64":
            assert self.core_prog is not None
            self.run(self.core_prog[self.target] if isinstance(self.core_prog, dict) else self.core_prog)
            self.calls.append(self.target)
            return
        if op == "s_mov_b32":
            self.s[a[0]] = self.rd(a[1])
            return
        if op == "s_mov_b64":
            if a[0] == "exec":
                self.exec = self.saved[a[1]]
            else:
                self.saved[a[0]] = self.exec
            return
        if op == "s_or_b64":
            self.c[a[0]] = self.c[a[1]] | self.c[a[2]]
            return
        if op == "s_andn2_b64" and a[0] != "exec":
            self.c[a[0]] = self.c[a[1]] & (1 - self.c[a[2]])
            self.trace.append(("salu", None, None))
            return
        if op == "s_andn2_b64":
            assert a[0] == "exec" and a[1] == "exec"
            self.exec = self.exec and not self.c[a[2]]
            return
        if op in ("ds_read_b128", "ds_write_b128", "global_load_dwordx4"):
            m = re.match(r"v\[(\d+):(\d+)\]", a[1] if op == "ds_write_b128" else a[0])
            regs = [f"v{k}" for k in range(int(m.group(1)), int(m.group(2)) + 1)]
            off = int(rest.split("offset:")[1]) if "offset:" in rest else 0
            if op == "ds_read_b128":
                base = self.rd(a[1].split()[0]) + off
                for k, r in enumerate(regs):
                    self.v[r] = ("pending", self.lds[base + 4 * k])
                self.loads_in_flight.append(("lds", regs))
            elif op == "global_load_dwordx4":
                sp = re.match(r"(s\[\d+:\d+\])", a[2]).group(1)
                base = self.rd(a[1]) + self.s64[sp] + off
                for k, r in enumerate(regs):
                    self.v[r] = ("pending", self.glob[base + 4 * k])
                self.loads_in_flight.append(("vm", regs))
            else:
                base = self.rd(a[0]) + off
                for k, r in enumerate(regs):
                    self.lds[base + 4 * k] = self.rd(r)
                self.store_reads.append((len(self.trace), regs))
            self.trace.append(("mem", None, None))
            return
        if op == "s_waitcnt":
            vm = int(re.search(r"vmcnt\((\d+)\)", rest).group(1)) if "vmcnt" in rest else None
            lgkm = int(re.search(r"lgkmcnt\((\d+)\)", rest).group(1)) if "lgkmcnt" in rest else None
            for kind, n in (("vm", vm), ("lds", lgkm)):
                if n is None:
                    continue
                mine = [x for x in self.loads_in_flight if x[0] == kind]
                done = mine[:max(0, len(mine) - n)]            # in-order return: all but the last n
                for x in done:
                    for r in x[1]:
                        if isinstance(self.v.get(r), tuple):
                            self.v[r] = self.v[r][1]
                    self.loads_in_flight.remove(x)
            self.trace.append(("salu", None, None))
            return
        if op == "v_add_u32_e32":
            self.wr(a[0], self.rd(a[1]) + self.rd(a[2]))
            self.trace.append(("valu", None, None))
            return
        # ---- VALU ----
        if op == "v_mov_b32_e32":
            self.wr(a[0], self.rd(a[1]))
        elif op == "v_mad_u64_u32":
            lo, hi = self.pair(a[0])
            add = 0 if a[4] == "0" else (self.rd(self.pair(a[4])[0]) | (self.rd(self.pair(a[4])[1]) << 32))
            r = self.rd(a[2]) * self.rd(a[3]) + add
            assert r < 1 << 64, "v_mad_u64_u32 overflow: the zero-high invariant is broken"
            self.wr(lo, r)
            self.wr(hi, r >> 32)
        elif op == "v_mul_lo_u32":
            self.wr(a[0], self.rd(a[1]) * self.rd(a[2]))
        elif op in ("v_add_co_u32_e32", "v_add_co_u32_e64", "v_sub_co_u32_e32", "v_sub_co_u32_e64"):
            x, y = self.rd(a[2]), self.rd(a[3])
            r = x + y if "add" in op else x - y
            self.wr(a[0], r)
            if self.exec:
                self.c[a[1]] = int(r > M32 or r < 0)
            carry_write = a[1]
        elif op in ("v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_subb_co_u32_e32", "v_subb_co_u32_e64"):
            x, y, cin = self.rd(a[2]), self.rd(a[3]), (self.c[a[4]] if self.exec else self.c.get(a[4], 0))   # a masked lane's carry is never used
            r = x + y + cin if "addc" in op else x - y - cin
            self.wr(a[0], r)
            if self.exec:
                self.c[a[1]] = int(r > M32 or r < 0)
            carry_read, carry_write = a[4], a[1]
        elif op in ("v_cndmask_b32_e64", "v_cndmask_b32_e32"):
            self.wr(a[0], self.rd(a[2]) if self.c[a[3]] else self.rd(a[1]))
            carry_read = a[3]
        elif op == "v_alignbit_b32":
            self.wr(a[0], (((self.rd(a[1]) << 32) | self.rd(a[2])) >> (self.rd(a[3]) & 31)))
        elif op == "v_lshlrev_b32_e32":
            self.wr(a[0], self.rd(a[2]) << (self.rd(a[1]) & 31))
        elif op == "v_or3_b32":
            self.wr(a[0], self.rd(a[1]) | self.rd(a[2]) | self.rd(a[3]))
        elif op == "v_or_b32_e32":
            self.wr(a[0], self.rd(a[1]) | self.rd(a[2]))
        elif op == "v_xor_b32_e32":
            self.wr(a[0], self.rd(a[1]) ^ self.rd(a[2]))
        elif op in ("v_cmp_eq_u32_e64", "v_cmp_ne_u32_e64"):
            eq = self.rd(a[1]) == self.rd(a[2])
            self.c[a[0]] = int(self.exec and (eq if "eq" in op else not eq))
            carry_write = a[0]
        else:
            raise AssertionError("instruction not modelled: " + ins)
        self.trace.append(("valu", carry_read, carry_write))

    def check_store_hold(self, hold=2):
        """no register an LDS store reads is rewritten within `hold` slots of the store"""
        for at, regs in self.store_reads:
            for slot, r in self.reg_writes:
                assert not (r in regs and at < slot <= at + hold), f"{r} rewritten {slot - at} slots after the store that reads it"

    def check_carry_hazard(self):
        last_write = {}
        for k, (kind, rd, wr) in enumerate(self.trace):
            if kind == "valu" and rd is not None and rd in last_write:
                assert k - last_write[rd] > G.CARRY_GAP, f"VALU reads {rd} {k - last_write[rd]} slots after its VALU write"
            if wr is not None:
                last_write[wr] = k


def limbs(x, n):
    return [(x >> (32 * i)) & M32 for i in range(n)]


def unlimbs(ws):
    return sum(w << (32 * i) for i, w in enumerate(ws))


CD_BASE = G.CD_BASE       # where the fused mixed addition keeps c and d of the two-product routine (its banks R3 and R2)


def run_mul(name, N, p, lazy, a, b, core=False, kind="mul", c=0, d=0):
    body, nv, res = G.function_text(name, N, p, lazy=lazy, core=core, kind=kind, cd_base=CD_BASE if kind == "mul2" else None)
    lane = Lane()
    if kind == "mul2":
        for j in range(N):
            lane.v[f"v{CD_BASE[0] + j}"] = limbs(c, N)[j]
            lane.v[f"v{CD_BASE[1] + j}"] = limbs(d, N)[j]
    if core:   # what the fused caller provides: modulus SGPRs, zero halves, T_N = 0
        inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
        for j in range(N):
            lane.s[f"s{36 + j}"] = (p >> (32 * j)) & M32
        lane.s[f"s{36 + N}"] = inv
        for j in range(N + 1):
            lane.v[f"v{2 * N + 2 * j + 1}"] = 0
        lane.v[f"v{2 * N + 2 * N}"] = 0
    for j in range(N):
        lane.v[f"v{j}"] = limbs(a, N)[j]
        lane.v[f"v{N + j}"] = limbs(b, N)[j]
    lane.run(body)
    lane.check_carry_hazard()
    return unlimbs([lane.v[f"v{r}"] for r in res]), lane


def operands(rng, p, hi):
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 255, (1 << 32) - 1]
    if hi > p:
        edge += [p, p + 1, 2 * p - 1]
    return [v for v in edge if v < hi] + [rng.randrange(hi) for _ in range(24)]


@pytest.mark.parametrize("name,N,p,lazy,core", [("fq", 12, G.Q, True, False), ("fq_core", 12, G.Q, True, True), ("fr", 8, G.R, False, False)])
def test_montgomery_product_model(name, N, p, lazy, core):
    rng = random.Random(100 + N + core)
    Rm = 1 << (32 * N)
    hi = 2 * p if lazy else p
    ops = operands(rng, p, hi)
    for a in ops:
        for b in (ops[rng.randrange(len(ops))], ops[rng.randrange(len(ops))], a):
            r, _ = run_mul(name, N, p, lazy, a, b, core)
            assert r % p == a * b * pow(Rm, -1, p) % p, (name, hex(a), hex(b))
            assert r < (2 * p if lazy else p), (name, "range", hex(a), hex(b))


@pytest.mark.parametrize("core", [False, True])
def test_montgomery_square_model(core):
    """the 78-product squaring: a * a * R^-1 in the lazy range, for every representative a < 2q; B registers are scratch"""
    N, p = 12, G.Q
    rng = random.Random(300 + core)
    Rm = 1 << (32 * N)
    extra = [2 * p - 1, 2 * p - 2, (1 << 381) - 1, (1 << 381), p + (1 << 380), 0x80000000, 0xFFFFFFFF << 32, int("80000000" * 11, 16)]
    for a in operands(rng, p, 2 * p) + [v for v in extra if v < 2 * p] + [rng.randrange(2 * p) for _ in range(40)]:
        junk = rng.randrange(1 << 384)                  # whatever the B registers held before
        r, lane = run_mul("sqr", N, p, True, a, junk, core, kind="sqr")
        assert r % p == a * a * pow(Rm, -1, p) % p, hex(a)
        assert r < 2 * p
        assert unlimbs([lane.v[f"v{j}"] for j in range(N)]) == a          # the operand survives


def test_montgomery_two_product_model():
    """a * b + c * d with one reduction: congruent to (ab + cd) R^-1 and below 2q for a, b, c, d < 2q and for c = 2q (the negated
    Y1 = 0 of the fused addition); T_N ends as 0 (what the next core call relies on)"""
    N, p = 12, G.Q
    rng = random.Random(411)
    Rm = 1 << (32 * N)
    ops = operands(rng, p, 2 * p)
    cases = [(2 * p - 1,) * 4, (2 * p - 1, 2 * p - 1, 2 * p, 2 * p - 1), (0, 0, 0, 0), (1, 1, 2 * p, 1), (0, 5, 2 * p, 2 * p - 1)]
    cases += [tuple(ops[rng.randrange(len(ops))] for _ in range(4)) for _ in range(40)]
    cases += [tuple(rng.randrange(2 * p) for _ in range(4)) for _ in range(40)]
    for a, b, c, d in cases:
        r, lane = run_mul("mul2", N, p, True, a, b, True, kind="mul2", c=c, d=d)
        assert r % p == (a * b + c * d) * pow(Rm, -1, p) % p, (hex(a), hex(b), hex(c), hex(d))
        assert r < 2 * p
        assert lane.v[f"v{2 * N + 2 * N}"] == 0


@pytest.mark.parametrize("sub", [True, False])
def test_add_sub_block_model(sub):
    N, p = 12, G.Q
    rng = random.Random(7 + sub)
    out = [f"v{100 + j}" for j in range(N)]
    ar = [f"v{j}" for j in range(N)]
    br = [f"v{20 + j}" for j in range(N)]
    D = [f"v{40 + j}" for j in range(N)]
    Pv = [f"v{60 + j}" for j in range(N)]
    L = []
    G.emit_addsub(L, N, sub, out, ar, br, D, Pv, "s[58:59]", 2 * p)
    ops = operands(rng, p, 2 * p)
    for a in ops:
        for b in ops[:12] + [a]:
            lane = Lane()
            for j in range(N):
                lane.v[ar[j]], lane.v[br[j]] = limbs(a, N)[j], limbs(b, N)[j]
            lane.run(L)
            lane.check_carry_hazard()
            r = unlimbs([lane.v[x] for x in out])
            assert r < 2 * p and r % p == ((a - b) if sub else (a + b)) % p


# ---- the fused mixed addition -------------------------------------------------------------------------------------
def xyzz_madd(p, X1, Y1, ZZ1, ZZZ1, x2, y2):
    """madd-2008-s on integers mod p (values are Montgomery residues: products carry R^-1)."""
    Ri = pow(1 << 384, -1, p)
    mul = lambda a, b: a * b * Ri % p
    U2, S2 = mul(x2, ZZ1), mul(y2, ZZZ1)
    P, Rr = (U2 - X1) % p, (S2 - Y1) % p
    if P == 0:
        return None
    PP = mul(P, P)
    PPP = mul(P, PP)
    Qv = mul(X1, PP)
    X3 = (mul(Rr, Rr) - PPP - 2 * Qv) % p
    Y3 = (mul(Rr, (Qv - X3) % p) - mul(Y1, PPP)) % p
    return X3, Y3, mul(ZZ1, PP), mul(ZZZ1, PPP)


def run_fused(acc, q, special, affine_acc=False, negy=0):
    N, p = 12, G.Q
    L, nv = G.fused_madd_program(p, affine_acc)
    cores = {"sonic_mont_mul_fq_core": G.function_text("core", N, p, lazy=True, core=True)[0],
             "sonic_mont_sqr_fq_core": G.function_text("core", N, p, lazy=True, core=True, kind="sqr")[0],
             "sonic_mont_mul2_fq_core": G.function_text("core", N, p, lazy=True, core=True, kind="mul2", cd_base=G.CD_BASE)[0]}
    lane = Lane(core_prog=cores)
    for k in range(4):
        for j in range(N):
            lane.v[f"%{k * N + j}"] = limbs(acc[k], N)[j]
    for k in range(2):
        for j in range(N):
            lane.v[f"%{4 * N + 1 + k * N + j}"] = limbs(q[k], N)[j]
    lane.v[f"%{6 * N + 1}"] = special
    lane.v[f"%{6 * N + 2}"] = negy
    lane.run(L)
    assert lane.exec is True, "EXEC not restored"
    lane.check_carry_hazard()
    assert sorted(lane.calls) == ["sonic_mont_mul2_fq_core"] + ["sonic_mont_mul_fq_core"] * (2 if affine_acc else 6) + ["sonic_mont_sqr_fq_core"] * 2
    out = [unlimbs([lane.v[f"%{k * N + j}"] for j in range(N)]) for k in range(4)]
    return out, lane.v[f"%{4 * N}"]


def test_fused_mixed_addition_model():
    p = G.Q
    rng = random.Random(2024)
    rep = lambda v: v + p if (rng.random() < 0.5 and v + p < 2 * p) else v          # either representative of the residue
    for it in range(16):
        acc = [rng.randrange(p) for _ in range(4)]
        q = [rng.randrange(p) for _ in range(2)]
        if it == 0:
            acc[1] = 0                               # Y1 = 0: the negated operand of the two-product core is 2q itself
        want = xyzz_madd(p, *acc, *q)
        got, exc = run_fused([rep(v) for v in acc], [rep(v) for v in q], 0)
        assert exc == 0 and all(g < 2 * p for g in got)
        assert [g % p for g in got] == list(want)
    # the sign of a signed digit: negy != 0 adds (q.x, -q.y), for either representative of q.y and for q.y = 0 (whose negation is the
    # representative 2q: the product takes it; such a point is of order two -- not on this curve's prime-order part, but the
    # statement must not care)
    for it in range(12):
        acc = [rng.randrange(p) for _ in range(4)]
        q = [rng.randrange(p) for _ in range(2)]
        if it == 0:
            q[1] = 0
        want = xyzz_madd(p, *acc, q[0], (-q[1]) % p)
        got, exc = run_fused([rep(v) for v in acc], [rep(q[0]), rep(q[1]) if q[1] else 0], 0, negy=1 + it)
        assert exc == 0 and all(g < 2 * p for g in got)
        assert [g % p for g in got] == list(want)
    # U2 == X1 (the doubling / cancellation position): flagged, accumulator untouched -- with U2 - X1 represented as 0 or as q
    Ri = pow(1 << 384, -1, p)
    for bump in (0, p):
        acc = [rng.randrange(p) for _ in range(4)]
        q = [rng.randrange(p) for _ in range(2)]
        acc[0] = q[0] * acc[2] * Ri % p + bump if q[0] * acc[2] * Ri % p + bump < 2 * p else q[0] * acc[2] * Ri % p
        got, exc = run_fused(acc, q, 0)
        assert exc == 1 and got == acc
    # an operand at infinity is announced by the caller: flagged, untouched
    acc = [rng.randrange(p) for _ in range(4)]
    q = [rng.randrange(p) for _ in range(2)]
    got, exc = run_fused(acc, q, 1)
    assert exc == 1 and got == acc


def test_fused_affine_plus_affine_model():
    """the variant for the second entry of a bucket walk: the accumulator comes in affine (ZZ = ZZZ = 1 implied, whatever acc.zz /
    acc.zzz held is ignored), 2 + 2 + 1 core calls; same results as the general formulas with ZZ1 = ZZZ1 = 1 (Montgomery: R mod q)"""
    p = G.Q
    rng = random.Random(4242)
    one = (1 << 384) % p
    rep = lambda v: v + p if (rng.random() < 0.5 and v + p < 2 * p) else v
    for it in range(16):
        x1, y1, x2, y2 = (rng.randrange(p) for _ in range(4))
        if it == 0:
            y1 = 0
        want = xyzz_madd(p, x1, y1, one, one, x2, y2)
        got, exc = run_fused([rep(x1), rep(y1), rng.randrange(1 << 384), rng.randrange(1 << 384)], [rep(x2), rep(y2)], 0, affine_acc=True)
        assert exc == 0 and all(g < 2 * p for g in got)
        assert [g % p for g in got] == list(want)
    for bump in (0, p):                                   # equal x (doubling / cancellation), with x2 - x1 represented as 0 or as q
        x1, y1, y2 = (rng.randrange(p) for _ in range(3))
        acc = [x1, y1, 7, 9]
        got, exc = run_fused(acc, [x1 + bump, y2], 0, affine_acc=True)
        assert exc == 1 and got == acc
    acc = [rng.randrange(p), rng.randrange(p), 1, 2]
    got, exc = run_fused(acc, [rng.randrange(p), rng.randrange(p)], 1, affine_acc=True)
    assert exc == 1 and got == acc


def test_committed_header_is_generated():
    """sonic_amd/csrc/mont_asm.hpp is what the generator produces today (nobody edited one without the other)"""
    path = os.path.join(os.path.dirname(__file__), "..", "sonic_amd", "csrc", "mont_asm.hpp")
    assert open(path).read() == G.render()


# ---- the NTT butterflies ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("U,hi", [(4, 16), (2, 16), (2, 16384), (2, 32768)])
@pytest.mark.parametrize("inverse,unit", [(False, False), (True, False), (False, True)])
def test_ntt_butterfly_model(inverse, unit, U, hi):
    """four butterflies of one stage out of LDS: operands anywhere in the lazy range [0, 2r) (edge values included), twiddles
    canonical; results congruent to the radix-2 butterfly, again below 2r, written where the operands came from; no operand is
    read before the wait that covers its load, no carry inside the hazard window, no store's data rewritten under it.
    hi = 16: elements as 32-byte records; hi = 16384 / 32768 (round 5): the split layout -- lower halves at a 16-byte stride, upper
    halves one plane further -- that removes the two-way bank conflict of the records"""
    N, p = 8, G.R
    Rm = 1 << 256
    Ri = pow(Rm, -1, p)
    rng = random.Random(77 + inverse)
    rg = G.ntt_regs(U)
    body = G.ntt_bfly_text(inverse, unit, U, hi)[:-1]         # without the return (unit: the stage whose twiddles are all 1)
    es = 32 if hi == 16 else 16                                # bytes between consecutive elements
    addr = lambda e, k: e + 4 * k if k < 4 else e + hi + 4 * (k - 4)     # noqa: E731  (limb k of the element at e)
    assert sum(1 for l in body if l.startswith("s_nop")) <= 8
    edge = [0, 1, p - 1, p, p + 1, 2 * p - 1, 2 * p - 2, (1 << 255), (1 << 255) - 1]
    for it in range(12):
        lane = Lane()
        span, twbase = es * rng.choice([1, 2, 64, 1024 if hi == 16 else 256]), 0x7F0012340000
        lane.s[G.NTT_SPAN] = span
        lane.s64[G.NTT_TWB] = twbase
        e0s, tws = [], []
        vals = []
        for u in range(U):
            e0 = (1024 if hi == 16 else 256) * it + (es * u if span >= es * U else 2 * span * u)        # the butterflies of a stage never share an element
            a = edge[(it + u) % len(edge)] if it < 6 else rng.randrange(2 * p)
            b = edge[(it * 3 + u + 1) % len(edge)] if it < 9 else rng.randrange(2 * p)
            w = Rm % p if unit else [0, 1, p - 1, Rm % p][(u + 2 * (U == 2)) % 4] if it == 0 else rng.randrange(p)
            t = 32 * rng.randrange(1 << 20)
            for k in range(N):
                lane.lds[addr(e0, k)], lane.lds[addr(e0 + span, k)], lane.glob[twbase + t + 4 * k] = limbs(a, N)[k], limbs(b, N)[k], limbs(w, N)[k]
            lane.v[f"v{rg['E0'] + u}"], lane.v[f"v{rg['TW'] + u}"] = e0, t
            e0s.append(e0); tws.append(t); vals.append((a, b, w))
        lane.run(body)
        lane.check_carry_hazard()
        lane.check_store_hold()
        assert not lane.loads_in_flight
        for u, (a, b, w) in enumerate(vals):
            x = unlimbs([lane.lds[addr(e0s[u], k)] for k in range(N)])
            y = unlimbs([lane.lds[addr(e0s[u] + span, k)] for k in range(N)])
            assert x < 2 * p and y < 2 * p, (it, u)
            if not inverse:
                assert x % p == (a + b) % p and y % p == (a - b) * w * Ri % p, (it, u, hex(a), hex(b), hex(w))
            else:
                assert x % p == (a + b * w * Ri) % p and y % p == (a - b * w * Ri) % p, (it, u, hex(a), hex(b), hex(w))
        # the zero halves and T_N are left as the next call expects them -- they are rewritten by the prologue anyway
