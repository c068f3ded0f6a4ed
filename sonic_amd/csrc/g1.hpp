// G1 of BLS12-381 (y^2 = x^3 + 4 over Fq): the group law the reference gets from
// elliptic-curve-0.3.0 (`<>`, `mul`, `mempty`, `gen`; call sites src/Sonic/CommitmentScheme.hs:26-29,
// 45-48 and src/Sonic/SRS.hs:33-39).
//
// Storage in HBM: affine (x, y), Montgomery form, 2 x 12 x u32 = 96 B, infinity = (0, 0)
// ((0,0) is not on the curve).  Accumulators: extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; infinity = ZZ == 0), 192 B -- the cheapest mixed addition (8M + 2S) and no
// inversion until the very end.  Every formula handles P+P, P+(-P) and infinity operands: they do
// occur (bench/Main.hs:23 uses x = 1, so all SRS points coincide).
#pragma once
#include "field.hpp"

namespace sonic {

struct G1Affine {
  Fq x, y;
  // infinity is only ever written as literal zeros (never the result of arithmetic), so the strict test is exact even in
  // the lazy range, where a coordinate congruent to 0 may be stored as q
  HD bool is_inf() const { return x.is_zero_strict() && y.is_zero_strict(); }
  static HD G1Affine inf() { G1Affine p; p.x = Fq::zero(); p.y = Fq::zero(); return p; }
};

// Affine points at a byte stride.  Caller-supplied point arrays are packed (96 B); the SRS bases and their window tables sit at
// SONIC_SRS_POINT_BYTES = 128: one HBM line per point.  A 96-B point at a 96-B stride straddles two 128-B lines half of the time, and
// the bucket walks gather table points at random: padded, a gather touches one line instead of 1.5 (measured on the walk itself,
// tools/ba_bench: 5.90 against 5.75 x 10^9 additions/s; HBM traffic of k_bucket_accum -1/3) for +33 % table memory -- HBM capacity
// (288 GB) is what this design spends.
#ifndef SONIC_SRS_POINT_BYTES
#define SONIC_SRS_POINT_BYTES 128
#endif
struct PointArray {
  const char* p;
  uint32_t stride;
  HD const G1Affine& operator[](size_t i) const { return *reinterpret_cast<const G1Affine*>(p + i * (size_t)stride); }
  HD PointArray operator+(long i) const { return PointArray{p + i * (long)stride, stride}; }
  static HD PointArray packed(const G1Affine* a) { return PointArray{reinterpret_cast<const char*>(a), (uint32_t)sizeof(G1Affine)}; }
};
struct PointArrayMut {
  char* p;
  uint32_t stride;
  HD G1Affine& operator[](size_t i) const { return *reinterpret_cast<G1Affine*>(p + i * (size_t)stride); }
  HD PointArrayMut operator+(long i) const { return PointArrayMut{p + i * (long)stride, stride}; }
  HD operator PointArray() const { return PointArray{p, stride}; }
};

struct G1XYZZ {
  Fq x, y, zz, zzz;
  HD bool is_inf() const { return zz.is_zero_strict(); }     // ZZ = 0 is written, never computed: ZZ3 = ZZ1 * PP with PP != 0
  static HD G1XYZZ inf() {
    G1XYZZ p; p.x = Fq::zero(); p.y = Fq::zero(); p.zz = Fq::zero(); p.zzz = Fq::zero(); return p;
  }
  static HD G1XYZZ from_affine(const G1Affine& a) {
    G1XYZZ p;
    if (a.is_inf()) return inf();
    p.x = a.x; p.y = a.y; p.zz = Fq::one(); p.zzz = Fq::one();
    return p;
  }
};

HD G1Affine g1_neg(const G1Affine& p) { G1Affine r; r.x = p.x; r.y = fp_neg(p.y); return r; }
HD G1XYZZ g1_neg(const G1XYZZ& p) { G1XYZZ r = p; r.y = fp_neg(p.y); return r; }

// 2 * (affine P) -> XYZZ  (mdbl-2008-s-1, a = 0)
HD G1XYZZ g1_dbl_affine(const G1Affine& p) {
  if (p.is_inf() || p.y.is_zero()) return G1XYZZ::inf();
  G1XYZZ r;
  Fq U = fp_dbl(p.y);
  Fq V = fp_sqr(U);
  Fq W = fp_mul(U, V);
  Fq S = fp_mul(p.x, V);
  Fq X2 = fp_sqr(p.x);
  Fq M = fp_add(fp_dbl(X2), X2);
  r.x = fp_sub(fp_sqr(M), fp_dbl(S));
  r.y = fp_sub(fp_mul(M, fp_sub(S, r.x)), fp_mul(W, p.y));
  r.zz = V;
  r.zzz = W;
  return r;
}

// 2 * (XYZZ P)  (dbl-2008-s-1, a = 0)
HD G1XYZZ g1_dbl(const G1XYZZ& p) {
  if (p.is_inf() || p.y.is_zero()) return G1XYZZ::inf();
  G1XYZZ r;
  Fq U = fp_dbl(p.y);
  Fq V = fp_sqr(U);
  Fq W = fp_mul(U, V);
  Fq S = fp_mul(p.x, V);
  Fq X2 = fp_sqr(p.x);
  Fq M = fp_add(fp_dbl(X2), X2);
  r.x = fp_sub(fp_sqr(M), fp_dbl(S));
  r.y = fp_sub(fp_mul(M, fp_sub(S, r.x)), fp_mul(W, p.y));
  r.zz = fp_mul(V, p.zz);
  r.zzz = fp_mul(W, p.zzz);
  return r;
}

// acc + (affine q)  (madd-2008-s): 8M + 2S on the generic path
HD G1XYZZ g1_add_mixed(const G1XYZZ& acc, const G1Affine& q) {
  if (q.is_inf()) return acc;
  if (acc.is_inf()) return G1XYZZ::from_affine(q);
  Fq U2 = fp_mul(q.x, acc.zz);
  Fq S2 = fp_mul(q.y, acc.zzz);
  Fq Pp = fp_sub(U2, acc.x);
  Fq R = fp_sub(S2, acc.y);
  if (Pp.is_zero()) {
    if (R.is_zero()) return g1_dbl_affine(q);
    return G1XYZZ::inf();
  }
  G1XYZZ r;
  Fq PP = fp_sqr(Pp);
  Fq PPP = fp_mul(Pp, PP);
  Fq Qv = fp_mul(acc.x, PP);
  r.x = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Qv));
  r.y = fp_sub(fp_mul(R, fp_sub(Qv, r.x)), fp_mul(acc.y, PPP));
  r.zz = fp_mul(acc.zz, PP);
  r.zzz = fp_mul(acc.zzz, PPP);
  return r;
}

// The same addition for the bucket walks: on the device the whole formula is one generated asm statement around ten calls of
// the product core (mont_asm.hpp, sonic_g1_madd_asm); lanes in an exceptional position come back flagged and unchanged and are
// redone by the general function above.
// negy != 0: acc - q (the sign of a signed digit: the fused statement negates q.y on the way into its first product -- 24
// instructions instead of the ~95 of a 12-limb negation in C++ ahead of it).
HD G1XYZZ g1_add_mixed_walk(G1XYZZ acc, const G1Affine& q, uint32_t negy = 0) {
#if SONIC_FQ_LAZY && !defined(SONIC_NO_FUSED_MADD)
  const uint32_t special = (acc.is_inf() || q.is_inf()) ? 1u : 0u;
  if (sonic_g1_madd_asm(acc, q.x, q.y, special, negy)) {
    G1Affine t = q;
    if (negy) t.y = fp_neg(t.y);
    acc = g1_add_mixed(acc, t);
  }
  return acc;
#else
  G1Affine t = q;
  if (negy) t.y = fp_neg(t.y);
  return g1_add_mixed(acc, t);
#endif
}

// first + q for two AFFINE points -> XYZZ: the second entry of a bucket walk.  With ZZ = ZZZ = 1 four of the ten products of the
// mixed addition are trivial; on the device the rest is one generated asm statement (sonic_g1_aadd_asm: 2 squarings, 2 products,
// 1 two-product call), exceptional lanes (an infinity operand, equal x) redone by the general functions.  `first_only`
// lanes just get from_affine(first) (a wave whose lanes differ in bucket size calls this for all of them).
HD G1XYZZ g1_add_affine_walk(const G1Affine& first, const G1Affine& q, bool first_only) {
#if SONIC_FQ_LAZY && !defined(SONIC_NO_FUSED_MADD)
  G1XYZZ acc;
  acc.x = first.x; acc.y = first.y; acc.zz = Fq::one(); acc.zzz = Fq::one();
  const uint32_t special = (first_only || first.is_inf() || q.is_inf()) ? 1u : 0u;
  if (sonic_g1_aadd_asm(acc, q.x, q.y, special)) {
    acc = G1XYZZ::from_affine(first);
    if (!first_only) acc = g1_add_mixed(acc, q);
  }
  return acc;
#else
  G1XYZZ acc = G1XYZZ::from_affine(first);
  return first_only ? acc : g1_add_mixed(acc, q);
#endif
}

// p + q, both XYZZ (add-2008-s): 12M + 2S
HD G1XYZZ g1_add(const G1XYZZ& p, const G1XYZZ& q) {
  if (q.is_inf()) return p;
  if (p.is_inf()) return q;
  Fq U1 = fp_mul(p.x, q.zz);
  Fq U2 = fp_mul(q.x, p.zz);
  Fq S1 = fp_mul(p.y, q.zzz);
  Fq S2 = fp_mul(q.y, p.zzz);
  Fq Pp = fp_sub(U2, U1);
  Fq R = fp_sub(S2, S1);
  if (Pp.is_zero()) {
    if (R.is_zero()) return g1_dbl(p);
    return G1XYZZ::inf();
  }
  G1XYZZ r;
  Fq PP = fp_sqr(Pp);
  Fq PPP = fp_mul(Pp, PP);
  Fq Qv = fp_mul(U1, PP);
  r.x = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Qv));
  r.y = fp_sub(fp_mul(R, fp_sub(Qv, r.x)), fp_mul(S1, PPP));
  r.zz = fp_mul(fp_mul(p.zz, q.zz), PP);
  r.zzz = fp_mul(fp_mul(p.zzz, q.zzz), PPP);
  return r;
}

// unique affine representative (one Fq inversion)
HD G1Affine g1_to_affine(const G1XYZZ& p) {
  if (p.is_inf()) return G1Affine::inf();
  Fq i = fp_inv(fp_mul(p.zz, p.zzz));
  G1Affine r;
  r.x = fp_mul(p.x, fp_mul(i, p.zzz));   // X / ZZ
  r.y = fp_mul(p.y, fp_mul(i, p.zz));    // Y / ZZZ
  return r;
}

// k * P for a small unsigned k (used by the bucket running-sum segments)
HD G1XYZZ g1_mul_small(const G1XYZZ& p, uint32_t k) {
  G1XYZZ acc = G1XYZZ::inf();
  bool started = false;
  for (int i = 31; i >= 0; i--) {
    if (started) acc = g1_dbl(acc);
    if ((k >> i) & 1) { acc = g1_add(acc, p); started = true; }
  }
  return acc;
}

}  // namespace sonic
