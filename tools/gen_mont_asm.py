#!/usr/bin/env python3
"""Generates sonic_amd/csrc/mont_asm.hpp: hand-scheduled gfx950 assembly for the Montgomery
product over Fq (12 x u32) and Fr (8 x u32), the fused point additions and the NTT butterflies built on it.

Why assembly: hipcc lowers the C++ CIOS loop to 288 v_mad_u64_u32 plus ~950 v_mov / v_lshl_add_u64
(it materialises every 64-bit addend), i.e. ~1250 VALU instructions per Fq product.  The routine
below needs 288 MADs + ~380 simple VALU instructions and no stack.

Algorithm (row-wise CIOS with "zero-high" accumulator pairs):
  the running value T (N+1 limbs) lives in the LOW halves of N+1 aligned VGPR pairs whose HIGH halves
  are permanently zero, so  Q_j = a_j * b_i + T_j  is a single v_mad_u64_u32 that cannot overflow.
  A row is: N independent MADs, one (N+1)-link carry chain that folds  sum Q_j 2^(32j)  back into T,
  m = T_0 * inv, N more MADs  R_j = m * p_j + T_j, and a second chain that folds and shifts down a limb.
  gfx950 needs two wait states between a VALU write of a carry (VCC / SGPR pair) and the VALU that
  consumes it, so a naive carry chain costs 3 issue slots per link; here the chains of consecutive
  phases run on different carry registers and are interleaved with the MADs that produce their inputs
  and consume their outputs (list scheduling below), so no s_nop is needed in the steady state.

Calling convention (private, not the C ABI): a in v[0:N), b in v[N:2N), result in v[0:N);
return address s[30:31]; clobbers v[2N : 6N+3), s[36:57], vcc, scc.  The C++ wrapper binds operands with generic "v"
constraints, moves them to / from the routine's fixed registers inside the asm text and calls with s_swappc_b64.
Fq runs in the lazy range [0, 2q) (no final conditional subtraction; result in the low halves of T); the "core" variant
leaves the prologue to its caller, the fused mixed addition sonic_g1_madd_asm (fused_madd_cxx below).

The same scheduler also builds the NTT butterfly routines (ntt_bfly_program): the two (or four) radix-2 butterflies a thread owns in
one stage as ONE program with the LDS reads, the twiddle loads and the LDS writes inside, Fr values in the lazy range [0, 2r).

Run from the repo root:  python tools/gen_mont_asm.py
"""
from __future__ import annotations

Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001

CARRY_GAP = 2          # instructions required between a VALU carry write and its VALU read
MAD_LATENCY = 4        # scheduling model only: slots before a v_mad_u64_u32 result is used
ALU_LATENCY = 1


class Ins:
    __slots__ = ("text", "reads", "writes", "idx", "deps", "users", "prio", "carry_reads", "hold")

    def __init__(self, text, reads, writes, carry_reads=()):
        self.text, self.reads, self.writes, self.carry_reads = text, set(reads), set(writes), set(carry_reads)
        self.deps, self.users, self.prio = set(), set(), 0
        self.hold = 0          # issue slots after this instruction in which nothing may overwrite what it reads (LDS / memory stores)


def build(N: int, p: int, lazy: bool = False, core: bool = False, kind: str = "mul", cd_base=None):
    """kind "mul": a * b.
    kind "sqr": a * a with 78 instead of 144 partial products: row i multiplies a_i by the limbs of
        a_i 2^(32 i) + 2 (a >> 32 (i+1)) 2^(32 (i+1)),  i.e. a_i itself at position i, (a_(i+1) << 1) at position i + 1 and the
        limbs x_j = (a_j << 1) | (a_(j-1) >> 31) of 2a above (the doubled high part differs from 2a only in the bit that the
        shift moves in from a_i); sum_i of these rows is a^2, every product is still 32 x 32 + 32 bits, and position k has all its
        terms by the time row k reduces it (they come from rows <= k / 2).  Needs a < 2^(32 N - 1) (lazy Fq: a < 2q < 2^382).
        The x_j live in the B registers (B is not an input); positions below i get no product in row i, so the first carry chain
        of row i starts at position i.
    kind "mul2": a * b + c * d with ONE reduction (c in v[cd_base[0] ...], d in v[cd_base[1] ...]): every row adds both
        products (two carry chains) before its reduction row.  With a, c <= 2q and b, d < 2q the running value stays below
        5q + epsilon < 2^383.1 (13 limbs suffice) and the result is < q (8 q / R + 1) < 1.82 q: inside the lazy range."""
    assert kind in ("mul", "sqr", "mul2") and (kind == "mul" or lazy)
    A = lambda j: j
    B = lambda j: N + j
    TPlo = lambda j: 2 * N + 2 * j
    TPhi = lambda j: 2 * N + 2 * j + 1
    QA = lambda j: 4 * N + 2 + 2 * j        # pair (lo, hi)
    QB = QA                                  # the reduction row reuses the product row's pairs (WAR tracked by the scheduler)
    M = 6 * N + 2
    SP = lambda j: 36 + j                    # modulus limbs in SGPRs
    SINV = 36 + N
    C1, C2, C3, JUNK = "vcc", "s[50:51]", "s[52:53]", "s[54:55]"
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)

    v = lambda r: f"v{r}"
    vp = lambda r: f"v[{r}:{r + 1}]"
    prog = []
    pre = [f"s_mov_b32 s{SP(j)}, 0x{(p >> (32 * j)) & 0xFFFFFFFF:08x}" for j in range(N)] + [f"s_mov_b32 s{SINV}, 0x{inv:08x}"]

    def emit(text, reads, writes, carry_reads=()):
        prog.append(Ins(text, reads, writes, carry_reads))

    def add_first(dst, c, x, y):            # dst = x + y, carry-out c
        if c == "vcc":
            emit(f"v_add_co_u32_e32 {v(dst)}, vcc, {v(x)}, {v(y)}", [v(x), v(y)], [v(dst), c])
        else:
            emit(f"v_add_co_u32_e64 {v(dst)}, {c}, {v(x)}, {v(y)}", [v(x), v(y)], [v(dst), c])

    def add_carry(dst, c, x, y):            # dst = x + y + c, carry-out c
        if c == "vcc":
            emit(f"v_addc_co_u32_e32 {v(dst)}, vcc, {v(x)}, {v(y)}, vcc", [v(x), v(y), c], [v(dst), c], [c])
        else:
            emit(f"v_addc_co_u32_e64 {v(dst)}, {c}, {v(x)}, {v(y)}, {c}", [v(x), v(y), c], [v(dst), c], [c])

    if not core:       # the "core" variant is called with the zero halves (and T_N = 0, which every product leaves behind) in place
        for j in range(N + 1):
            emit(f"v_mov_b32_e32 {v(TPhi(j))}, 0", [], [v(TPhi(j))])
        emit(f"v_mov_b32_e32 {v(TPlo(N))}, 0", [], [v(TPlo(N))])

    Cr = (lambda j: cd_base[0] + j) if kind == "mul2" else None
    Dr = (lambda j: cd_base[1] + j) if kind == "mul2" else None
    X = B                                    # "sqr": limbs of 2a (j >= 2); X(0) / X(1) alternate as the (a_(i+1) << 1) temporary
    if kind == "sqr":
        for j in range(2, N):
            emit(f"v_alignbit_b32 {v(X(j))}, {v(A(j))}, {v(A(j - 1))}, 31", [v(A(j)), v(A(j - 1))], [v(X(j))])

    def product_row(i, first, xs, mult, lo=0):
        """Q_j = xs[j] * mult + T_j for j in [lo, N) (addend 0 when `first`), then the carry chain that folds the Q pairs back
        into T: T_lo = lo(Q_lo), T_j = lo(Q_j) + hi(Q_(j-1)) + c, T_N += hi(Q_(N-1)) + c."""
        for j in range(lo, N):
            if first:
                emit(f"v_mad_u64_u32 {vp(QA(j))}, {JUNK}, {v(xs[j])}, {v(mult)}, 0", [v(xs[j]), v(mult)], [v(QA(j)), v(QA(j) + 1)])
            else:
                emit(f"v_mad_u64_u32 {vp(QA(j))}, {JUNK}, {v(xs[j])}, {v(mult)}, {vp(TPlo(j))}",
                     [v(xs[j]), v(mult), v(TPlo(j)), v(TPhi(j))], [v(QA(j)), v(QA(j) + 1)])

    def fold_chain(c, lo=0):
        emit(f"v_mov_b32_e32 {v(TPlo(lo))}, {v(QA(lo))}", [v(QA(lo))], [v(TPlo(lo))])
        started = False
        for j in range(lo + 1, N):
            (add_carry if started else add_first)(TPlo(j), c, QA(j), QA(j - 1) + 1)
            started = True
        (add_carry if started else add_first)(TPlo(N), c, TPlo(N), QA(N - 1) + 1)

    for i in range(N):
        if kind == "sqr":
            xs = {i: A(i)}
            if i + 1 < N:
                y = X(i & 1)
                emit(f"v_lshlrev_b32_e32 {v(y)}, 1, {v(A(i + 1))}", [v(A(i + 1))], [v(y)])
                xs[i + 1] = y
            for j in range(i + 2, N):
                xs[j] = X(j)
            product_row(i, i == 0, xs, A(i), lo=i)
            # position 0 of row i > 0 got no product: m comes from T_0 as the previous reduction row left it
            msrc = QA(0) if i == 0 else TPlo(0)
            emit(f"v_mul_lo_u32 {v(M)}, {v(msrc)}, s{SINV}", [v(msrc)], [v(M)])
            fold_chain(C1, lo=i)
        elif kind == "mul2":
            product_row(i, i == 0, [A(j) for j in range(N)], B(i))
            fold_chain(C1)
            product_row(i, False, [Cr(j) for j in range(N)], Dr(i))
            emit(f"v_mul_lo_u32 {v(M)}, {v(QA(0))}, s{SINV}", [v(QA(0))], [v(M)])
            fold_chain(C3)
        else:
            # multiplication row: Q_j = a_j * b_i + T_j
            product_row(i, i == 0, [A(j) for j in range(N)], B(i))
            emit(f"v_mul_lo_u32 {v(M)}, {v(QA(0))}, s{SINV}", [v(QA(0))], [v(M)])
            fold_chain(C1)
        # reduction row: R_j = m * p_j + T_j, then shift down one limb
        for j in range(N):
            emit(f"v_mad_u64_u32 {vp(QB(j))}, {JUNK}, {v(M)}, s{SP(j)}, {vp(TPlo(j))}",
                 [v(M), v(TPlo(j)), v(TPhi(j))], [v(QB(j)), v(QB(j) + 1)])
        c = C2
        for j in range(1, N):
            (add_first if j == 1 else add_carry)(TPlo(j - 1), c, QB(j), QB(j - 1) + 1)
        add_carry(TPlo(N - 1), c, TPlo(N), QB(N - 1) + 1)
        emit(f"v_addc_co_u32_e64 {v(TPlo(N))}, {c}, 0, 0, {c}", [c], [v(TPlo(N)), c], [c])

    if lazy:
        # "almost Montgomery": with 4p < R the CIOS result of operands < 2p is again < 2p (T_N == 0), so values simply live
        # in [0, 2p) and the conditional subtraction disappears; the caller reads the result from the low halves of T
        return pre, prog, 6 * N + 3, [TPlo(j) for j in range(N)]
    # final conditional subtraction: D = T - p; result = borrow ? T : D   (T < 2p, so T_N == 0)
    PV = lambda j: QA(0) + j                 # modulus limbs in VGPRs (the Q pairs are dead now)
    D = lambda j: QA(0) + N + j
    for j in range(N):
        emit(f"v_mov_b32_e32 {v(PV(j))}, s{SP(j)}", [], [v(PV(j))])
    for j in range(N):
        if j == 0:
            emit(f"v_sub_co_u32_e64 {v(D(j))}, {C3}, {v(TPlo(j))}, {v(PV(j))}", [v(TPlo(j)), v(PV(j))], [v(D(j)), C3])
        else:
            emit(f"v_subb_co_u32_e64 {v(D(j))}, {C3}, {v(TPlo(j))}, {v(PV(j))}, {C3}", [v(TPlo(j)), v(PV(j)), C3], [v(D(j)), C3], [C3])
    for j in range(N):
        emit(f"v_cndmask_b32_e64 {v(A(j))}, {v(D(j))}, {v(TPlo(j))}, {C3}", [v(D(j)), v(TPlo(j)), C3], [v(A(j))], [C3])
    return pre, prog, 6 * N + 3, [A(j) for j in range(N)]


def schedule(prog):
    """Dependence-respecting list scheduling with the carry-read wait-state constraint."""
    last_write, readers = {}, {}
    for k, ins in enumerate(prog):
        ins.idx = k
        for r in ins.reads:
            if r in last_write:
                ins.deps.add(last_write[r])
        for w in ins.writes:
            if w in last_write:
                ins.deps.add(last_write[w])           # WAW
            for rd in readers.get(w, ()):
                if rd != k:
                    ins.deps.add(rd)                 # WAR
        for r in ins.reads:
            readers.setdefault(r, []).append(k)
        for w in ins.writes:
            last_write[w] = k
            readers[w] = []
    for ins in prog:
        for d in ins.deps:
            prog[d].users.add(ins.idx)
    for ins in reversed(prog):                        # critical-path priority
        cost = 3 if ins.text.startswith("v_mad_u64") else 1
        ins.prio = cost + max((prog[u].prio for u in ins.users), default=0)
    done_at, out, remaining = {}, [], set(range(len(prog)))
    carry_written_at = {}
    slot = 0
    indeg = {k: len(prog[k].deps) for k in remaining}
    ready = {k for k in remaining if indeg[k] == 0}
    while remaining:
        best, best_key = None, None
        for k in ready:
            ins = prog[k]
            if any(slot - carry_written_at.get(c, -10) <= CARRY_GAP for c in ins.carry_reads):
                continue
            if any(prog[d].hold and slot - done_at[d] <= prog[d].hold for d in ins.deps):
                continue
            # latency model: prefer instructions whose producers' results have had time to land (a MAD result
            # takes a few issue slots); among those the longest remaining critical path
            lag = max((done_at[d] + (MAD_LATENCY if prog[d].text.startswith("v_mad_u64") else ALU_LATENCY) - slot
                       for d in ins.deps if d in done_at), default=0)
            key = (-(lag if lag > 0 else 0), ins.prio, -k)
            if best is None or key > best_key:
                best, best_key = k, key
        if best is None:
            out.append("s_nop 0")
            slot += 1
            continue
        ins = prog[best]
        out.append(ins.text)
        for w in ins.writes:
            if w in ("vcc",) or w.startswith("s["):
                carry_written_at[w] = slot
        done_at[best] = slot
        slot += 1
        ready.discard(best)
        remaining.discard(best)
        for u in ins.users:
            indeg[u] -= 1
            if indeg[u] == 0:
                ready.add(u)
    return out


def function_text(name: str, N: int, p: int, lazy: bool = False, core: bool = False, kind: str = "mul", cd_base=None):
    pre, prog, nv, res = build(N, p, lazy, core, kind, cd_base)
    body = ([] if core else pre) + schedule(prog) + ["s_setpc_b64 s[30:31]"]
    return body, nv, res


def routine_section(name: str, body, comment: str):
    """An out-of-line routine emitted from inside an asm statement into its own text section (see cxx())."""
    NL = "\\n\\t"
    lines = [comment, f'extern "C" __device__ __attribute__((noinline, used)) void {name}_holder() {{', "  asm volatile(",
             f'      ".pushsection .text.{name},\\"ax\\",@progbits{NL}"', f'      ".p2align 8{NL}"', f'      ".type {name},@function{NL}"',
             f'      "{name}:{NL}"']
    for l in body:
        lines.append(f'      "{l}{NL}"')
    lines += [f'      ".size {name}, .-{name}{NL}"', f'      ".popsection{NL}"', "  );", "}"]
    return lines


def limb32(v: int, j: int) -> int:
    return (v >> (32 * j)) & 0xFFFFFFFF


def emit_addsub(L, N, sub, out, a, b, D, Pv, CA, p2):
    """out = a -/+ b in the lazy range, same two-chain pattern as addsub_cxx, on explicit register names (strings)."""
    opA0, opA = ("v_sub_co_u32_e64", "v_subb_co_u32_e64") if sub else ("v_add_co_u32_e64", "v_addc_co_u32_e64")
    opB0, opB = ("v_add_co_u32_e32", "v_addc_co_u32_e32") if sub else ("v_sub_co_u32_e32", "v_subb_co_u32_e32")
    for j in range(N):
        L.append(f"v_mov_b32_e32 {Pv[j]}, 0x{limb32(p2, j):08x}")
        if j == 0:
            L.append(f"{opA0} {D[0]}, {CA}, {a[0]}, {b[0]}")
        else:
            L.append(f"{opA} {D[j]}, {CA}, {a[j]}, {b[j]}, {CA}")
        if j == 1:
            L.append(f"{opB0} {out[0]}, vcc, {D[0]}, {Pv[0]}")
        elif j >= 2:
            L.append(f"{opB} {out[j - 1]}, vcc, {D[j - 1]}, {Pv[j - 1]}, vcc")
        else:
            L.append("s_nop 0")
    L.append("s_nop 1")
    L.append(f"{opB} {out[N - 1]}, vcc, {D[N - 1]}, {Pv[N - 1]}, vcc")
    L.append("s_nop 1")
    for j in range(N):
        if sub:
            L.append(f"v_cndmask_b32_e64 {out[j]}, {D[j]}, {out[j]}, {CA}")
        else:
            L.append(f"v_cndmask_b32_e32 {out[j]}, {out[j]}, {D[j]}, vcc")


def emit_negate(L, N, out, a, const, Pv, fillers):
    """out = const - a (const: python integer): one borrow chain in VCC.  A literal and VCC cannot both feed one VOP2 (one
    constant-bus read), so the limbs of const above the first go through the scratch registers Pv; those moves and `fillers`
    (instructions that touch neither VCC nor out / a / Pv) are spent on the two slots the hardware wants between a carry write
    and its read, s_nop when they run out."""
    f = [f"v_mov_b32_e32 {Pv[j]}, 0x{limb32(const, j):08x}" for j in range(1, N)] + list(fillers)
    for j in range(N):
        if j == 0:
            L.append(f"v_sub_co_u32_e32 {out[0]}, vcc, 0x{limb32(const, 0):08x}, {a[0]}")
        else:
            L.append(f"v_subb_co_u32_e32 {out[j]}, vcc, {Pv[j]}, {a[j]}, vcc")
        if j < N - 1:
            for _ in range(2):
                L.append(f.pop(0) if f else "s_nop 0")
    L.extend(f)


CD_BASE = (99, 87)       # c / d operands of the two-product core: the fused addition's banks R3 and R2


def fused_madd_program(p: int, affine_acc: bool = False):
    """acc += q (XYZZ + affine, madd-2008-s) as ONE asm statement around ten calls of the product core.
    What it saves against ten separate product calls from C++ (per addition): the 25 zero-half initialisations of nine products
    (the zero halves and the modulus SGPRs persist across the calls), ~200 of the ~450 marshalling / merge moves (operands are
    routed between the core's fixed registers and four temporary banks by plan: the sub / add blocks write straight into the
    core's A operand or into the accumulator's registers), and every compiler-made copy.
    Lanes in an exceptional position (an infinity operand, or U2 = X1: doubling / cancellation) are switched off with EXEC before
    anything is written back and reported in `exc`: the caller redoes them with the general C++ addition.
    affine_acc: the accumulator comes in as an AFFINE point (x, y in acc.x / acc.y; ZZ = ZZZ = 1 implied, acc.zz / acc.zzz are
    outputs only) -- the second entry of a bucket walk.  U2 = x2, S2 = y2, ZZ3 = PP and ZZZ3 = PPP need no product:
    2 squarings + 2 products + 1 two-product call instead of 2 + 6 + 1."""
    N = 12
    A = [f"v{j}" for j in range(N)]
    B = [f"v{N + j}" for j in range(N)]
    T = [f"v{2 * N + 2 * j}" for j in range(N)]
    TPhi = [f"v{2 * N + 2 * j + 1}" for j in range(N + 1)]
    TPloN = f"v{2 * N + 2 * N}"
    SCR = [f"v{4 * N + 2 + j}" for j in range(2 * N)]          # the Q pairs: free between products
    D, Pv = SCR[:N], SCR[N:]
    base = 6 * N + 3                                           # 75
    R1 = [f"v{base + j}" for j in range(N)]                    # R
    R2 = [f"v{base + N + j}" for j in range(N)]                # PPP
    R3 = [f"v{base + 2 * N + j}" for j in range(N)]            # Q
    R4 = [f"v{base + 3 * N + j}" for j in range(N)]            # R * (Q - X3)
    nv = base + 4 * N
    ACCX = [f"%{j}" for j in range(N)]
    ACCY = [f"%{N + j}" for j in range(N)]
    ACCZZ = [f"%{2 * N + j}" for j in range(N)]
    ACCZZZ = [f"%{3 * N + j}" for j in range(N)]
    EXC = f"%{4 * N}"
    QX = [f"%{4 * N + 1 + j}" for j in range(N)]
    QY = [f"%{5 * N + 1 + j}" for j in range(N)]
    SPECIAL = f"%{6 * N + 1}"
    NEGY = f"%{6 * N + 2}"                                     # (mixed addition only) != 0: add (q.x, -q.y)
    CA, SAVE, MASK, TMPM, NEGM = "s[58:59]", "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]"
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    L = []
    mov = lambda dst, src: [L.append(f"v_mov_b32_e32 {dst[j]}, {src[j]}") for j in range(N)]

    def call(kind="mul"):
        L.append("s_getpc_b64 s[56:57]")
        L.append(f"s_add_u32 s56, s56, sonic_mont_{kind}_fq_core@rel32@lo+4")
        L.append(f"s_addc_u32 s57, s57, sonic_mont_{kind}_fq_core@rel32@hi+12")
        L.append("s_swappc_b64 s[30:31], s[56:57]")

    # prologue: modulus limbs / -p^-1 in SGPRs, zero halves of the T pairs, T_N = 0
    for j in range(N):
        L.append(f"s_mov_b32 s{36 + j}, 0x{limb32(p, j):08x}")
    L.append(f"s_mov_b32 s{36 + N}, 0x{inv:08x}")
    for r in TPhi:
        L.append(f"v_mov_b32_e32 {r}, 0")
    L.append(f"v_mov_b32_e32 {TPloN}, 0")
    L.append(f"s_mov_b64 {SAVE}, exec")
    if affine_acc:
        # 1. R = q.y - Y1;  2. P = q.x - X1 (into A), exceptional lanes off
        emit_addsub(L, N, True, R1, QY, ACCY, D, Pv, CA, 2 * p)
        emit_addsub(L, N, True, A, QX, ACCX, D, Pv, CA, 2 * p)
    else:
        # 1. S2 = (+-q.y) * ZZZ1;  R = S2 - Y1.  The sign of a signed digit is applied here, on the way into the product's operand:
        #    2q - q.y (one borrow chain; the copy of ZZZ1 fills its wait states) or q.y itself.  q.y = 0 gives 2q, which the product
        #    takes like any other representative (its bound needs a <= 2q: the result stays below 1.41 q) -- and such a point is at
        #    infinity or of order two and comes in flagged `special` anyway.
        L.append(f"v_cmp_ne_u32_e64 {NEGM}, 0, {NEGY}")
        emit_negate(L, N, D, QY, 2 * p, Pv, [f"v_mov_b32_e32 {B[j]}, {ACCZZZ[j]}" for j in range(N)])
        for j in range(N):
            L.append(f"v_cndmask_b32_e64 {A[j]}, {QY[j]}, {D[j]}, {NEGM}")
        call()
        emit_addsub(L, N, True, R1, T, ACCY, D, Pv, CA, 2 * p)
        # 2. U2 = q.x * ZZ1;  P = U2 - X1 (into A), exceptional lanes off
        mov(A, QX); mov(B, ACCZZ); call()
        emit_addsub(L, N, True, A, T, ACCX, D, Pv, CA, 2 * p)
    #    P == 0 (as 0 or as p)?  t = OR limbs, u = OR (limb ^ p_j)
    L.append(f"v_or3_b32 {D[0]}, {A[0]}, {A[1]}, {A[2]}")
    for k in range(3, N, 2):
        L.append(f"v_or3_b32 {D[0]}, {D[0]}, {A[k]}, {A[k + 1]}" if k + 1 < N else f"v_or_b32_e32 {D[0]}, {D[0]}, {A[k]}")
    for j in range(N):
        L.append(f"v_xor_b32_e32 {Pv[j]}, 0x{limb32(p, j):08x}, {A[j]}")
    L.append(f"v_or3_b32 {D[1]}, {Pv[0]}, {Pv[1]}, {Pv[2]}")
    for k in range(3, N, 2):
        L.append(f"v_or3_b32 {D[1]}, {D[1]}, {Pv[k]}, {Pv[k + 1]}" if k + 1 < N else f"v_or_b32_e32 {D[1]}, {D[1]}, {Pv[k]}")
    L.append(f"v_cmp_eq_u32_e64 {MASK}, 0, {D[0]}")
    L.append(f"v_cmp_eq_u32_e64 {TMPM}, 0, {D[1]}")
    L.append("s_nop 4")
    L.append(f"s_or_b64 {MASK}, {MASK}, {TMPM}")
    L.append(f"v_cmp_ne_u32_e64 {TMPM}, 0, {SPECIAL}")
    L.append("s_nop 4")
    L.append(f"s_or_b64 {MASK}, {MASK}, {TMPM}")
    L.append("s_nop 4")
    L.append(f"v_cndmask_b32_e64 {EXC}, 0, 1, {MASK}")
    L.append(f"s_andn2_b64 exec, exec, {MASK}")
    # 3. PP = P^2  (the squaring core: 78 partial products; clobbers B, keeps A)
    call("sqr")
    mov(B, T)
    if affine_acc:
        mov(ACCZZ, T)                       # ZZ3 = PP
    # 4. PPP = P * PP
    call()
    mov(R2, T)
    if affine_acc:
        mov(ACCZZZ, T)                      # ZZZ3 = PPP
    # 5. Q = X1 * PP
    mov(A, ACCX); call()
    mov(R3, T)
    if not affine_acc:
        # 6. ZZ3 = ZZ1 * PP
        mov(A, ACCZZ); call()
        mov(ACCZZ, T)
        # 7. ZZZ3 = ZZZ1 * PPP
        mov(A, ACCZZZ); mov(B, R2); call()
        mov(ACCZZZ, T)
    # 8. X3 = R^2 - PPP - 2 Q  (into the accumulator);  Q - X3 (into A)
    mov(A, R1); call("sqr")
    emit_addsub(L, N, True, R4, T, R2, D, Pv, CA, 2 * p)            # R^2 - PPP       (R4 is free from here on)
    emit_addsub(L, N, False, B, R3, R3, D, Pv, CA, 2 * p)           # 2 Q             (B is rewritten in step 9)
    emit_addsub(L, N, True, ACCX, R4, B, D, Pv, CA, 2 * p)          # X3
    emit_addsub(L, N, True, A, R3, ACCX, D, Pv, CA, 2 * p)          # Q - X3
    # 9. Y3 = R (Q - X3) + (2q - Y1) PPP: both products under ONE reduction (the two-product core reads c = 2q - Y1 from bank R3,
    #    whose Q is spent, and d = PPP where it already lives, bank R2)
    assert [f"v{CD_BASE[0] + j}" for j in range(N)] == R3 and [f"v{CD_BASE[1] + j}" for j in range(N)] == R2
    emit_negate(L, N, R3, ACCY, 2 * p, Pv, [f"v_mov_b32_e32 {B[j]}, {R1[j]}" for j in range(N)])
    call("mul2")
    mov(ACCY, T)
    L.append(f"s_mov_b64 exec, {SAVE}")
    return L, nv


def fused_madd_cxx(p: int, affine_acc: bool = False) -> str:
    """The C++ wrapper around fused_madd_program: operands %0..%47 = accumulator (in/out), %48 = exceptional flag (out),
    %49..%72 = the affine point, %73 = "an operand is at infinity" (in), %74 (mixed addition only) = "subtract the point" (in)."""
    N = 12
    L, nv = fused_madd_program(p, affine_acc)
    fname = "sonic_g1_aadd_asm" if affine_acc else "sonic_g1_madd_asm"
    ncalls = (2, 2, 1) if affine_acc else (6, 2, 1)
    n_mov = sum(1 for l in L if l.startswith("v_mov_b32"))
    NLs = "\\n\\t"
    outs = [f'"+v"(acc.x.l[{j}])' for j in range(N)] + [f'"+v"(acc.y.l[{j}])' for j in range(N)] + \
           [f'"+v"(acc.zz.l[{j}])' for j in range(N)] + [f'"+v"(acc.zzz.l[{j}])' for j in range(N)] + ['"=&v"(exc)']
    ins = [f'"v"(qx.l[{j}])' for j in range(N)] + [f'"v"(qy.l[{j}])' for j in range(N)] + ['"v"(special)'] + ([] if affine_acc else ['"v"(negy)'])
    clob = [f'"v{k}"' for k in range(nv)] + [f'"s{k}"' for k in [30, 31] + list(range(36, 68))] + ['"vcc"', '"scc"']
    head = [f"// {fname}: {len(L)} instructions around {ncalls[0]} calls of sonic_mont_mul_fq_core, {ncalls[1]} of sonic_mont_sqr_fq_core and {ncalls[2]} of "
            f"sonic_mont_mul2_fq_core ({n_mov} v_mov), VGPRs v0..v{nv - 1}" + ("; the accumulator comes in affine (acc.x, acc.y), ZZ = ZZZ = 1 implied" if affine_acc else ""),
            f"template <class XYZZ, class F> __device__ __forceinline__ bool {fname}(XYZZ& acc, const F& qx, const F& qy, uint32_t special" + ("" if affine_acc else ", uint32_t negy") + ") {",
            "  uint32_t exc;", "  asm volatile("]
    body = [f'      "{l}{NLs}"' for l in L[:-1]] + [f'      "{L[-1]}"']
    tail = [f"      : {', '.join(outs)}", f"      : {', '.join(ins)}", f"      : {', '.join(clob)});", "  return exc != 0;", "}"]
    return "\n".join(head + body + tail)



def cxx(name: str, cls: str, N: int, p: int, lazy: bool = False, kind: str = "mul") -> str:
    body, nv, res = function_text(name, N, p, lazy, kind=kind)
    nin = 1 if kind == "sqr" else 2
    nops = sum(1 for l in body if l.startswith("s_nop"))
    mads = sum(1 for l in body if l.startswith("v_mad_u64"))
    lines = [f"// {name}: {len(body)} instructions ({mads} v_mad_u64_u32, {nops} s_nop), VGPRs v0..v{nv - 1}"]
    # The routine is emitted from inside an asm statement into its own text section, not as an LLVM
    # function: a function (even a naked one) gets "s_waitcnt vmcnt(0) lgkmcnt(0)" at its entry, which would
    # drain the caller's prefetched point loads at every product.
    lines.append(f'extern "C" __device__ __attribute__((noinline, used)) void {name}_holder() {{')
    lines.append("  asm volatile(")
    NL = "\\n\\t"
    lines.append(f'      ".pushsection .text.{name},\\"ax\\",@progbits{NL}"')
    lines.append(f'      ".p2align 8{NL}"')
    lines.append(f'      ".type {name},@function{NL}"')
    lines.append(f'      "{name}:{NL}"')
    for l in body:
        lines.append(f'      "{l}{NL}"')
    lines.append(f'      ".size {name}, .-{name}{NL}"')
    lines.append(f'      ".popsection{NL}"')
    lines.append("  );")
    lines.append("}")
    # Operands use generic "v" constraints and are moved to / from the routine's fixed registers INSIDE the asm
    # text.  (Binding them with physical-register constraints, "{v0}" ..., made ROCm 7.2's clang silently drop
    # copies between consecutive calls -- wrong products in some kernels -- and crash its scheduler in others.)
    outs = ", ".join(f'"=v"(r.l[{j}])' for j in range(N))
    ins = ", ".join([f'"v"(a.l[{j}])' for j in range(N)] + ([f'"v"(b.l[{j}])' for j in range(N)] if nin == 2 else []))
    clob = [f'"v{k}"' for k in range(0, nv)] + [f'"s{k}"' for k in [30, 31] + list(range(36, 58))] + ['"vcc"', '"scc"']
    lines.append(f"__device__ __forceinline__ {cls} {name}_call(const {cls}& a" + (f", const {cls}& b" if nin == 2 else "") + ") {")
    lines.append(f"  {cls} r;")
    lines.append("  asm volatile(")
    for j in range(nin * N):
        lines.append(f'      "v_mov_b32_e32 v{j}, %{N + j}\\n\\t"')
    lines.append('      "s_getpc_b64 s[56:57]\\n\\t"')
    lines.append(f'      "s_add_u32 s56, s56, {name}@rel32@lo+4\\n\\t"')
    lines.append(f'      "s_addc_u32 s57, s57, {name}@rel32@hi+12\\n\\t"')
    lines.append('      "s_swappc_b64 s[30:31], s[56:57]\\n\\t"')
    for j in range(N):
        lines.append(f'      "v_mov_b32_e32 %{j}, v{res[j]}' + ('\\n\\t"' if j < N - 1 else '"'))
    lines.append(f"      : {outs}")
    lines.append(f"      : {ins}")
    lines.append(f"      : {', '.join(clob)});")
    lines.append("  return r;")
    lines.append("}")
    return "\n".join(lines)


def addsub_cxx(fname: str, cls: str, N: int, sub: bool, const: str = "p2") -> str:
    """r = a -/+ b as one asm block: two interleaved carry chains and a select.  Operands and result live in [0, 2p)
    (see the lazy Montgomery product), so the correction constant is 2p:
    sub:  d = a - b (borrow chain A), e = d + 2p (carry chain B, one link behind), r = borrow ? e : d
    add:  d = a + b (carry chain A),  e = d - 2p (borrow chain B),                 r = borrow(e) ? d : e
    Chain A carries in an SGPR pair, chain B in VCC; the v_mov that loads p_j is the filler that keeps two
    instructions between a carry write and its read (gfx950 VALU-carry hazard).
    const = "p": the same block for values kept canonical (Fr: operands and result in [0, p), the correction constant is p)."""
    opA0, opA = ("v_sub_co_u32_e64", "v_subb_co_u32_e64") if sub else ("v_add_co_u32_e64", "v_addc_co_u32_e64")
    opB0, opB = ("v_add_co_u32_e32", "v_addc_co_u32_e32") if sub else ("v_sub_co_u32_e32", "v_subb_co_u32_e32")
    # operand numbering: outputs r[0..N) = %0.., d[0..N) early-clobber temps, pv temps, cA (sgpr pair), then inputs a, b, p literals via "s"
    R_ = lambda j: f"%{j}"
    D_ = lambda j: f"%{N + j}"
    P_ = lambda j: f"%{2 * N + j}"
    CA = f"%{3 * N}"
    A_ = lambda j: f"%{3 * N + 1 + j}"
    B_ = lambda j: f"%{4 * N + 1 + j}"
    K_ = lambda j: f"%{5 * N + 1 + j}"         # modulus limb constants ("s": SGPR holding the literal)
    L = []
    for j in range(N):
        L.append(f"v_mov_b32_e32 {P_(j)}, {K_(j)}")
        if j == 0:
            L.append(f"{opA0} {D_(0)}, {CA}, {A_(0)}, {B_(0)}")
        else:
            L.append(f"{opA} {D_(j)}, {CA}, {A_(j)}, {B_(j)}, {CA}")
        if j == 1:
            L.append(f"{opB0} {R_(0)}, vcc, {D_(0)}, {P_(0)}")
        elif j >= 2:
            L.append(f"{opB} {R_(j - 1)}, vcc, {D_(j - 1)}, {P_(j - 1)}, vcc")
        else:
            L.append("s_nop 0")
    # tail: last link of chain B, then selects
    L.append("s_nop 1")
    L.append(f"{opB} {R_(N - 1)}, vcc, {D_(N - 1)}, {P_(N - 1)}, vcc" if N > 1 else f"{opB0} {R_(0)}, vcc, {D_(0)}, {P_(0)}")
    L.append("s_nop 1")
    for j in range(N):
        if sub:   # borrow of chain A set -> take e (already in r), else d
            L.append(f"v_cndmask_b32_e64 {R_(j)}, {D_(j)}, {R_(j)}, {CA}")
        else:     # add: carry-out of A cannot happen (a + b < 2^(32N)); borrow of chain B (vcc) set -> d < p -> take d
            L.append(f"v_cndmask_b32_e32 {R_(j)}, {R_(j)}, {D_(j)}, vcc")
    outs = [f'"=&v"(r.l[{j}])' for j in range(N)] + [f'"=&v"(d{j})' for j in range(N)] + [f'"=&v"(p{j})' for j in range(N)] + ['"=&s"(ca)']
    ins = [f'"v"(a.l[{j}])' for j in range(N)] + [f'"v"(b.l[{j}])' for j in range(N)] + [f'"s"(P::{const}({j}))' for j in range(N)]
    body = "\\n\\t".join(L)
    decl = " ".join(f"uint32_t d{j}, p{j};" for j in range(N))
    return f"""template <class P> __device__ __forceinline__ {cls} {fname}(const {cls}& a, const {cls}& b) {{
  {cls} r; {decl} unsigned long long ca;
  asm volatile("{body}"
      : {', '.join(outs)}
      : {', '.join(ins)}
      : "vcc");
  return r;
}}"""


# ---- NTT butterflies over Fr: the U butterflies a thread owns in one stage, operands in LDS, as ONE scheduled program ----------
NTT_U = 4
NTT_SPAN, NTT_TWB = "s70", "s[68:69]"


def ntt_regs(U: int = NTT_U):
    """VGPR map of the butterfly routines: T pairs v0..v17, Q pairs v18..v33, m v34, limbs of 2r v36..v43, butterfly u: a v[44+24u ..],
    b (+8), w (+16), then the sum x (8; the other results go where an operand was) and -- four butterflies per thread -- two banks of
    chain temporaries (8 + 8).  The two-butterfly routines (four waves per SIMD: at most 128 VGPRs for the whole kernel) borrow the Q
    pairs for those: their chains then wait for the products instead of filling slots between them.  Inputs: LDS byte address of the
    butterfly's first element (E0 + u), byte offset of its twiddle (TW + u); the second element's address goes to E1 + u."""
    top = 44 + 24 * U
    if U >= 4:
        x, d1, d2, nxt = top, top + 8, top + 16, top + 24
    else:
        x, d1, d2, nxt = top, 18, 26, top + 8
    return {"X": x, "D1": d1, "D2": d2, "E0": nxt, "TW": nxt + U, "E1": nxt + 2 * U, "NV": nxt + 3 * U}


NTT_NV = ntt_regs()["NV"]            # VGPRs v0 .. v175
NTT_E0, NTT_TW, NTT_E1 = ntt_regs()["E0"], ntt_regs()["TW"], ntt_regs()["E1"]


def ntt_bfly_program(inverse: bool, U: int = NTT_U, unit: bool = False, hi: int = 16):
    """hi: byte offset of an element's upper four limbs from its lower four in LDS.  16 = the element as one 32-byte record.  With 32-byte
    records a ds_read_b128 of a wave touches every OTHER 16-byte slot -- banks 0-3, 8-11, ... -- a two-way bank conflict on every LDS access
    of the transform (rocprofv3: SQ_LDS_BANK_CONFLICT = 63-68 % of SQ_LDS_IDX_ACTIVE in k_ntt_wide / k_ntt_local, profiles/r05_lds_*).  The
    split layout (round 5) keeps the lower halves of a block's elements in one plane (element i at byte 16 i) and the upper halves in a
    second plane `hi` bytes further: consecutive lanes then read consecutive 16-byte slots.  `hi` is an immediate of the ds instructions
    (<= 65535), so every plane distance has its own routines: 16384 for 1024-element blocks, 32768 for 2048-element tiles.

    One stage's butterflies of a thread, values in the lazy range [0, 2r) (2r < 2^256), twiddles canonical:
        forward (DIF):  x = a + b,      y = (a - b) w        inverse (DIT):  t = b w,  x = a + t,  y = a - t
    a = LDS[e0], b = LDS[e0 + span], w = table[tw]; x and y go back where a and b came from.  A product of a value below 2r with a
    canonical twiddle is below r (2r / 2^256 + 1) < 1.91 r without any final subtraction, sums and differences are brought back
    below 2r by one conditional -+ 2r.  All loads are issued first; the products run one after another (they share the T / Q
    registers), the add / sub chains fill the slots around them.
    unit: the stage whose twiddles are all 1 (span of one element): x = a + b, y = a - b in either direction, no loads from the
    table, no product.
    Returns (prologue, scheduled body, epilogue) as instruction lists."""
    N, p = 8, R
    p2 = 2 * p
    assert p2 < 1 << 256
    TPlo = lambda j: 2 * j
    TPhi = lambda j: 2 * j + 1
    QA = lambda j: 18 + 2 * j
    M = 34
    P2 = lambda j: 36 + j
    Ar = lambda u, j: 44 + 24 * u + j
    Br = lambda u, j: 52 + 24 * u + j
    Wr = lambda u, j: 60 + 24 * u + j
    rg = ntt_regs(U)
    X = lambda j: rg["X"] + j
    D1 = lambda j: rg["D1"] + j
    D2 = lambda j: rg["D2"] + j
    NTT_E0, NTT_TW, NTT_E1 = rg["E0"], rg["TW"], rg["E1"]
    assert Wr(U - 1, 7) < rg["X"]
    SP = lambda j: 36 + j
    SINV = 36 + N
    C1, C2, JUNK = "vcc", "s[50:51]", "s[54:55]"
    CA1, CB1, CA2, CB2, MASK = "s[58:59]", "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]"
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    v = lambda r: f"v{r}"
    vp = lambda r: f"v[{r}:{r + 1}]"
    v4 = lambda r: f"v[{r}:{r + 3}]"

    pre = [] if unit else [f"s_mov_b32 s{SP(j)}, 0x{limb32(p, j):08x}" for j in range(N)] + [f"s_mov_b32 s{SINV}, 0x{inv:08x}"]
    for u in range(U):
        pre.append(f"v_add_u32_e32 {v(NTT_E1 + u)}, {NTT_SPAN}, {v(NTT_E0 + u)}")
        pre.append(f"ds_read_b128 {v4(Ar(u, 0))}, {v(NTT_E0 + u)}")
        pre.append(f"ds_read_b128 {v4(Ar(u, 4))}, {v(NTT_E0 + u)} offset:{hi}")
        pre.append(f"ds_read_b128 {v4(Br(u, 0))}, {v(NTT_E1 + u)}")
        pre.append(f"ds_read_b128 {v4(Br(u, 4))}, {v(NTT_E1 + u)} offset:{hi}")
        if not unit:
            pre.append(f"global_load_dwordx4 {v4(Wr(u, 0))}, {v(NTT_TW + u)}, {NTT_TWB}")
            pre.append(f"global_load_dwordx4 {v4(Wr(u, 4))}, {v(NTT_TW + u)}, {NTT_TWB} offset:16")
    if not unit:
        for j in range(N + 1):
            pre.append(f"v_mov_b32_e32 {v(TPhi(j))}, 0")
        pre.append(f"v_mov_b32_e32 {v(TPlo(N))}, 0")
    for j in range(N):
        pre.append(f"v_mov_b32_e32 {v(P2(j))}, 0x{limb32(p2, j):08x}")

    prog = []

    def emit(text, reads, writes, carry_reads=()):
        prog.append(Ins(text, reads, writes, carry_reads))
        return prog[-1]

    def chain(op0, op, dst, c, x, y):
        """dst = x op y limb by limb, carry / borrow in the SGPR pair (or vcc) c"""
        for j in range(N):
            if j == 0:
                emit(f"{op0}_e64 {v(dst(j))}, {c}, {v(x(j))}, {v(y(j))}", [v(x(j)), v(y(j))], [v(dst(j)), c])
            else:
                emit(f"{op}_e64 {v(dst(j))}, {c}, {v(x(j))}, {v(y(j))}, {c}", [v(x(j)), v(y(j)), c], [v(dst(j)), c], [c])

    def add_lazy(out, x, y, D, ca, cb):
        """out = x + y (- 2r if that is not negative).  x + y may pass 2^256: with the carry c1 of the sum and the borrow b1 of the
        subtraction as a ninth limb, (x + y - 2r) is negative exactly when b1 and not c1."""
        chain("v_add_co_u32", "v_addc_co_u32", D, ca, x, y)
        chain("v_sub_co_u32", "v_subb_co_u32", out, cb, D, P2)
        emit(f"s_andn2_b64 {MASK}, {cb}, {ca}", [cb, ca], [MASK], [cb, ca])
        for j in range(N):
            emit(f"v_cndmask_b32_e64 {v(out(j))}, {v(out(j))}, {v(D(j))}, {MASK}", [v(out(j)), v(D(j)), MASK], [v(out(j))])

    def sub_lazy(out, x, y, D, ca, cb):
        """out = x - y (+ 2r if the difference is negative)"""
        chain("v_sub_co_u32", "v_subb_co_u32", D, ca, x, y)
        chain("v_add_co_u32", "v_addc_co_u32", out, cb, D, P2)
        for j in range(N):
            emit(f"v_cndmask_b32_e64 {v(out(j))}, {v(D(j))}, {v(out(j))}, {ca}", [v(D(j)), v(out(j)), ca], [v(out(j))], [ca])

    def add_first(dst, c, x, y):
        if c == "vcc":
            emit(f"v_add_co_u32_e32 {v(dst)}, vcc, {v(x)}, {v(y)}", [v(x), v(y)], [v(dst), c])
        else:
            emit(f"v_add_co_u32_e64 {v(dst)}, {c}, {v(x)}, {v(y)}", [v(x), v(y)], [v(dst), c])

    def add_carry(dst, c, x, y):
        if c == "vcc":
            emit(f"v_addc_co_u32_e32 {v(dst)}, vcc, {v(x)}, {v(y)}, vcc", [v(x), v(y), c], [v(dst), c], [c])
        else:
            emit(f"v_addc_co_u32_e64 {v(dst)}, {c}, {v(x)}, {v(y)}, {c}", [v(x), v(y), c], [v(dst), c], [c])

    def product(xa, xb, out):
        """out (eight consecutive registers) = xa * xb / 2^256 mod r without the final subtraction; same rows as build(kind="mul"),
        the last reduction row folds into `out` instead of the T pairs.  Returns the first instruction (see the waits below)."""
        head = None
        for i in range(N):
            for j in range(N):
                if i == 0:
                    ins = emit(f"v_mad_u64_u32 {vp(QA(j))}, {JUNK}, {v(xa(j))}, {v(xb(i))}, 0", [v(xa(j)), v(xb(i))], [v(QA(j)), v(QA(j) + 1)])
                    head = head or ins
                else:
                    emit(f"v_mad_u64_u32 {vp(QA(j))}, {JUNK}, {v(xa(j))}, {v(xb(i))}, {vp(TPlo(j))}",
                         [v(xa(j)), v(xb(i)), v(TPlo(j)), v(TPhi(j))], [v(QA(j)), v(QA(j) + 1)])
            emit(f"v_mul_lo_u32 {v(M)}, {v(QA(0))}, s{SINV}", [v(QA(0))], [v(M)])
            emit(f"v_mov_b32_e32 {v(TPlo(0))}, {v(QA(0))}", [v(QA(0))], [v(TPlo(0))])
            for j in range(1, N):
                (add_first if j == 1 else add_carry)(TPlo(j), C1, QA(j), QA(j - 1) + 1)
            add_carry(TPlo(N), C1, TPlo(N), QA(N - 1) + 1)
            for j in range(N):
                emit(f"v_mad_u64_u32 {vp(QA(j))}, {JUNK}, {v(M)}, s{SP(j)}, {vp(TPlo(j))}",
                     [v(M), v(TPlo(j)), v(TPhi(j))], [v(QA(j)), v(QA(j) + 1)])
            dst = (lambda j: out(j)) if i == N - 1 else TPlo
            for j in range(1, N):
                (add_first if j == 1 else add_carry)(dst(j - 1), C2, QA(j), QA(j - 1) + 1)
            add_carry(dst(N - 1), C2, TPlo(N), QA(N - 1) + 1)
            emit(f"v_addc_co_u32_e64 {v(TPlo(N))}, {C2}, 0, 0, {C2}", [C2], [v(TPlo(N)), C2], [C2])
        return head

    def store(addr, src):
        for h in (0, 4):
            ins = emit(f"ds_write_b128 {v(addr)}, {v4(src(h))}" + (f" offset:{hi}" if h else ""), [v(addr)] + [v(src(h + k)) for k in range(4)], [])
            ins.hold = 2          # its data registers must not be rewritten in the next two slots

    # Waits: the LDS reads all at once (a partial lgkmcnt would have to trust that nothing else is in flight on that counter), the
    # twiddles one butterfly at a time, each behind the product of the butterfly before (by then it has long arrived).  The forward
    # butterflies run all their sums and differences first -- those need the LDS operands only, the twiddles are still in flight.
    A_ = lambda u: (lambda j: Ar(u, j))
    B_ = lambda u: (lambda j: Br(u, j))
    W_ = lambda u: (lambda j: Wr(u, j))
    lds_regs = [v(Ar(k, j)) for k in range(U) for j in range(N)] + [v(Br(k, j)) for k in range(U) for j in range(N)]
    emit("s_waitcnt lgkmcnt(0)", lds_regs, lds_regs)

    def wait_twiddle(u, after):
        regs = [v(Wr(u, j)) for j in range(N)]
        emit(f"s_waitcnt vmcnt({2 * (U - 1 - u)})", regs + after, regs)

    if unit:
        for u in range(U):
            add_lazy(W_(u), A_(u), B_(u), D1, CA1, CB1)           # (the twiddle registers are free: every butterfly has its own outputs)
            store(NTT_E0 + u, W_(u))
            sub_lazy(A_(u), A_(u), B_(u), D2, CA2, CB2)
            store(NTT_E1 + u, A_(u))
    elif not inverse:
        for u in range(U):
            add_lazy(X, A_(u), B_(u), D1, CA1, CB1)
            store(NTT_E0 + u, X)
            sub_lazy(A_(u), A_(u), B_(u), D2, CA2, CB2)           # a - b where a was
        for u in range(U):
            wait_twiddle(u, [v(Ar(k, j)) for k in range(U) for j in range(N)] if u == 0 else [v(QA(0))])
            product(A_(u), W_(u), B_(u))                          # y where b was
            store(NTT_E1 + u, B_(u))
    else:
        for u in range(U):
            wait_twiddle(u, [] if u == 0 else [v(QA(0))])
            product(B_(u), W_(u), W_(u))                          # t where w was
            add_lazy(X, A_(u), W_(u), D1, CA1, CB1)
            store(NTT_E0 + u, X)
            sub_lazy(B_(u), A_(u), W_(u), D2, CA2, CB2)           # y where b was
            store(NTT_E1 + u, B_(u))
    post = ["s_waitcnt lgkmcnt(0)"]
    return pre, prog, post


def ntt_bfly_text(inverse: bool, unit: bool = False, U: int = NTT_U, hi: int = 16):
    pre, prog, post = ntt_bfly_program(inverse, U, unit=unit, hi=hi)
    return pre + schedule(prog) + post + ["s_setpc_b64 s[30:31]"]


def ntt_bfly_cxx(inverse: bool, unit: bool = False, U: int = NTT_U, hi: int = 16) -> str:
    assert hi == 16 or (hi & (hi - 1)) == 0 and 16 < hi <= 32768
    name = f"sonic_ntt_bfly{U}" + ("" if hi == 16 else f"s{hi.bit_length() - 1}") + "_" + ("unit" if unit else "inv" if inverse else "fwd")
    body = ntt_bfly_text(inverse, unit, U, hi)
    rg = ntt_regs(U)
    NTT_E0, NTT_TW, NTT_NV = rg["E0"], rg["TW"], rg["NV"]
    mads = sum(1 for l in body if l.startswith("v_mad_u64"))
    nops = sum(1 for l in body if l.startswith("s_nop"))
    what = "x = a + b, y = a - b: the stage whose twiddles are all 1" if unit else "t = b w, x = a + t, y = a - t" if inverse else "x = a + b, y = (a - b) w"
    layout = "32-byte records" if hi == 16 else f"split layout: lower halves at 16-byte stride, upper halves {hi} bytes further"
    lines = routine_section(name, body, f"// {name}: {U} radix-2 butterflies ({what}) on Fr values in LDS ({layout}), lazy range [0, 2r): {len(body)} instructions "
                                        f"({mads} v_mad_u64_u32, {nops} s_nop), VGPRs v0..v{NTT_NV - 1}")
    args = ", ".join([f"uint32_t e{u}" for u in range(U)] + ([] if unit else [f"uint32_t t{u}" for u in range(U)]) + ["uint32_t span"] + ([] if unit else ["const void* twiddles"]))
    lines.append("// e_u: LDS byte address of butterfly u's first element (the second one is `span` bytes further)" + ("" if unit else ", t_u: byte offset of its twiddle from `twiddles`"))
    lines.append(f"__device__ __forceinline__ void {name}({args}) {{")
    lines.append("  asm volatile(")
    k = 0
    for u in range(U):
        lines.append(f'      "v_mov_b32_e32 v{NTT_E0 + u}, %{k}\\n\\t"'); k += 1
    if not unit:
        for u in range(U):
            lines.append(f'      "v_mov_b32_e32 v{NTT_TW + u}, %{k}\\n\\t"'); k += 1
    lines.append(f'      "s_mov_b32 {NTT_SPAN}, %{k}\\n\\t"'); k += 1
    if not unit:
        lines.append(f'      "s_mov_b64 {NTT_TWB}, %{k}\\n\\t"'); k += 1
    lines.append('      "s_getpc_b64 s[56:57]\\n\\t"')
    lines.append(f'      "s_add_u32 s56, s56, {name}@rel32@lo+4\\n\\t"')
    lines.append(f'      "s_addc_u32 s57, s57, {name}@rel32@hi+12\\n\\t"')
    lines.append('      "s_swappc_b64 s[30:31], s[56:57]"')
    ins = ", ".join([f'"v"(e{u})' for u in range(U)] + ([] if unit else [f'"v"(t{u})' for u in range(U)]) + ['"s"(span)'] + ([] if unit else ['"s"(twiddles)']))
    clob = [f'"v{k}"' for k in range(NTT_NV)] + [f'"s{k}"' for k in [30, 31] + list(range(36, 71))] + ['"vcc"', '"scc"', '"memory"']
    lines.append("      :")
    lines.append(f"      : {ins}")
    lines.append(f"      : {', '.join(clob)});")
    lines.append("}")
    return "\n".join(lines)


def render() -> str:
    """The text of sonic_amd/csrc/mont_asm.hpp (tests/test_asm_model.py checks that the committed file is this)."""
    out = ["// GENERATED by tools/gen_mont_asm.py -- do not edit.",
           "// Hand-scheduled gfx950 Montgomery products (see the generator for the algorithm and the schedule).",
           "#pragma once",
           "#if defined(__HIP_DEVICE_COMPILE__)",
           "namespace sonic {",
           cxx("sonic_mont_mul_fq", "Fp<FqParams>", 12, Q, lazy=True),
           "",
           cxx("sonic_mont_sqr_fq", "Fp<FqParams>", 12, Q, lazy=True, kind="sqr"),
           "",
           cxx("sonic_mont_mul_fr", "Fp<FrParams>", 8, R),
           "",
           "\n".join(routine_section("sonic_mont_mul_fq_core", function_text("sonic_mont_mul_fq_core", 12, Q, lazy=True, core=True)[0],
                                     "// sonic_mont_mul_fq_core: the lazy Fq product without its prologue (modulus SGPRs, zero halves): callers keep those in place")),
           "",
           "\n".join(routine_section("sonic_mont_sqr_fq_core", function_text("sonic_mont_sqr_fq_core", 12, Q, lazy=True, core=True, kind="sqr")[0],
                                     "// sonic_mont_sqr_fq_core: a * a with 78 partial products (a in v0..v11 survives, v12..v23 are scratch), same contract as the core above")),
           "",
           "\n".join(routine_section("sonic_mont_mul2_fq_core", function_text("sonic_mont_mul2_fq_core", 12, Q, lazy=True, core=True, kind="mul2", cd_base=CD_BASE)[0],
                                     f"// sonic_mont_mul2_fq_core: a * b + c * d under one reduction (c in v{CD_BASE[0]}.., d in v{CD_BASE[1]}..), same contract as the core above")),
           "",
           fused_madd_cxx(Q),
           "",
           fused_madd_cxx(Q, affine_acc=True),
           "",
           addsub_cxx("sonic_fq_sub_asm", "Fp<P>", 12, True),
           addsub_cxx("sonic_fq_add_asm", "Fp<P>", 12, False),
           addsub_cxx("sonic_fr_sub_asm", "Fp<P>", 8, True, const="p"),
           addsub_cxx("sonic_fr_add_asm", "Fp<P>", 8, False, const="p"),
           "",
           ntt_bfly_cxx(False),
           "",
           ntt_bfly_cxx(True),
           "",
           ntt_bfly_cxx(False, unit=True),
           "",
           ntt_bfly_cxx(False, U=2),
           "",
           ntt_bfly_cxx(True, U=2),
           "",
           ntt_bfly_cxx(False, unit=True, U=2),
           "",
           "// the same two-butterfly routines over the split LDS layout (no two-way bank conflict): 1024-element blocks / 2048-element tiles",
           ntt_bfly_cxx(False, U=2, hi=16384),
           "",
           ntt_bfly_cxx(True, U=2, hi=16384),
           "",
           ntt_bfly_cxx(False, unit=True, U=2, hi=16384),
           "",
           ntt_bfly_cxx(False, U=2, hi=32768),
           "",
           ntt_bfly_cxx(True, U=2, hi=32768),
           "",
           ntt_bfly_cxx(False, unit=True, U=2, hi=32768),
           "}  // namespace sonic",
           "#endif", ""]
    return "\n".join(out)


def main():
    open("sonic_amd/csrc/mont_asm.hpp", "w").write(render())
    for name, N, p in (("fq", 12, Q), ("fr", 8, R)):
        body, nv, _ = function_text(name, N, p, lazy=(name == "fq"))
        print(name, "instructions:", len(body), "nops:", sum(1 for l in body if l.startswith("s_nop")), "vgprs:", nv)


if __name__ == "__main__":
    main()
