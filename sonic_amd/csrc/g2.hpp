// G2 of BLS12-381: the twist y^2 = x^3 + 4(u + 1) over Fq2 = Fq[u]/(u^2 + 1).
// Only SRS.new touches G2 (src/Sonic/SRS.hs:35-36,40-41: hNegativeX, hPositiveX, h*AlphaX); the prover never
// reads these vectors, the verifier reads three elements.  Jacobian accumulators (3 x Fq2) keep the register
// footprint below the XYZZ form's.
#pragma once
#include "g1.hpp"

namespace sonic {

struct Fq2 {
  Fq c0, c1;
  static HD Fq2 zero() { Fq2 r; r.c0 = Fq::zero(); r.c1 = Fq::zero(); return r; }
  static HD Fq2 one() { Fq2 r; r.c0 = Fq::one(); r.c1 = Fq::zero(); return r; }
  HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  HD bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
};
HD Fq2 f2_add(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
HD Fq2 f2_sub(const Fq2& a, const Fq2& b) { Fq2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
HD Fq2 f2_dbl(const Fq2& a) { return f2_add(a, a); }
HD Fq2 f2_neg(const Fq2& a) { Fq2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + ((a0 + a1)(b0 + b1) - a0 b0 - a1 b1) u
HD Fq2 f2_mul(const Fq2& a, const Fq2& b) {
  Fq t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
  Fq t2 = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
  Fq2 r;
  r.c0 = fp_sub(t0, t1);
  r.c1 = fp_sub(fp_sub(t2, t0), t1);
  return r;
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u
HD Fq2 f2_sqr(const Fq2& a) {
  Fq2 r;
  r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
  r.c1 = fp_dbl(fp_mul(a.c0, a.c1));
  return r;
}
HD Fq2 f2_inv(const Fq2& a) {
  Fq d = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  Fq2 r;
  r.c0 = fp_mul(a.c0, d);
  r.c1 = fp_neg(fp_mul(a.c1, d));
  return r;
}

struct G2Affine {
  Fq2 x, y;
  HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  static HD G2Affine inf() { G2Affine p; p.x = Fq2::zero(); p.y = Fq2::zero(); return p; }
};
struct G2Jac {
  Fq2 x, y, z;
  HD bool is_inf() const { return z.is_zero(); }
  static HD G2Jac inf() { G2Jac p; p.x = Fq2::one(); p.y = Fq2::one(); p.z = Fq2::zero(); return p; }
};

HD G2Jac g2_dbl(const G2Jac& p) {           // dbl-2009-l, a = 0
  if (p.is_inf() || p.y.is_zero()) return G2Jac::inf();
  Fq2 A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
  Fq2 t = f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C);
  Fq2 D = f2_dbl(t);
  Fq2 E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
  G2Jac r;
  r.x = f2_sub(F, f2_dbl(D));
  Fq2 C8 = f2_dbl(f2_dbl(f2_dbl(C)));
  r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), C8);
  r.z = f2_dbl(f2_mul(p.y, p.z));
  return r;
}
HD G2Jac g2_add_mixed(const G2Jac& p, const G2Affine& q) {   // madd-2007-bl
  if (q.is_inf()) return p;
  if (p.is_inf()) { G2Jac r; r.x = q.x; r.y = q.y; r.z = Fq2::one(); return r; }
  Fq2 Z1Z1 = f2_sqr(p.z), U2 = f2_mul(q.x, Z1Z1), S2 = f2_mul(f2_mul(q.y, p.z), Z1Z1);
  if (U2 == p.x) {
    if (S2 == p.y) return g2_dbl(p);
    return G2Jac::inf();
  }
  Fq2 H = f2_sub(U2, p.x), HH = f2_sqr(H), I = f2_dbl(f2_dbl(HH)), J = f2_mul(H, I);
  Fq2 rr = f2_dbl(f2_sub(S2, p.y)), V = f2_mul(p.x, I);
  G2Jac r;
  r.x = f2_sub(f2_sub(f2_sqr(rr), J), f2_dbl(V));
  r.y = f2_sub(f2_mul(rr, f2_sub(V, r.x)), f2_dbl(f2_mul(p.y, J)));
  r.z = f2_sub(f2_sub(f2_sqr(f2_add(p.z, H)), Z1Z1), HH);
  return r;
}
HD G2Affine g2_to_affine(const G2Jac& p) {
  if (p.is_inf()) return G2Affine::inf();
  Fq2 zi = f2_inv(p.z), zi2 = f2_sqr(zi);
  G2Affine r;
  r.x = f2_mul(p.x, zi2);
  r.y = f2_mul(p.y, f2_mul(zi2, zi));
  return r;
}

}  // namespace sonic
