// The pairing of the verifier (host only): pcV's equation eA <> eB == eC (src/Sonic/CommitmentScheme.hs:58-68) as a product of
// three Miller loops and one final exponentiation.
//
// The reference takes `pairing` from pairing-1.0.0; only equalities of pairing products are ever tested, so any bilinear,
// non-degenerate map on (G1, G2) accepts exactly the same proofs.  Used here: the ate loop over |x| = 0xd201000000010000 with the
// running point kept on the twist E': y^2 = x^3 + 4(1 + u) in Jacobian coordinates, lines evaluated at the G1 point as sparse
// elements of Fq12 = Fq2[w]/(w^6 - xi), xi = 1 + u (the untwisting map is (x, y) -> (x / w^2, y / w^3)), and the exponent
// 3 (q^12 - 1) / r = (q^6 - 1)(q^2 + 1) * [ (x - 1)^2 (x + q)(x^2 + q^2 - 1) + 3 ]   (identity checked in tests/test_pairing_host.py).
// Lines are scaled by elements of Fq2 and by w^3 (in Fq4): both die in the factor q^4 - 1 of the exponent.
//
// tests/pairing_selftest.cpp compiles this header with g++ and checks it against the plain polynomial-basis pairing the verifier
// used before (tests/pairing_plain.hpp: affine arithmetic in Fq12, exponent (q^12 - 1)/r bit by bit): value == plain^3.
#pragma once
#include "g2.hpp"
#include "constants.hpp"

namespace sonic {
namespace pairing {

inline Fq2 f2_conj(const Fq2& a) { Fq2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
inline Fq2 f2_mul_xi(const Fq2& a) { Fq2 r; r.c0 = fp_sub(a.c0, a.c1); r.c1 = fp_add(a.c0, a.c1); return r; }   // (a0 + a1 u)(1 + u)
inline Fq2 f2_mul_fq(const Fq2& a, const Fq& s) { Fq2 r; r.c0 = fp_mul(a.c0, s); r.c1 = fp_mul(a.c1, s); return r; }

// Fq6 = Fq2[v] / (v^3 - xi)
struct F6 {
  Fq2 a0, a1, a2;
  static F6 zero() { F6 r; r.a0 = r.a1 = r.a2 = Fq2::zero(); return r; }
  static F6 one() { F6 r = zero(); r.a0 = Fq2::one(); return r; }
  bool is_zero() const { return a0.is_zero() && a1.is_zero() && a2.is_zero(); }
  bool operator==(const F6& o) const { return a0 == o.a0 && a1 == o.a1 && a2 == o.a2; }
};
inline F6 f6_add(const F6& a, const F6& b) { F6 r; r.a0 = f2_add(a.a0, b.a0); r.a1 = f2_add(a.a1, b.a1); r.a2 = f2_add(a.a2, b.a2); return r; }
inline F6 f6_sub(const F6& a, const F6& b) { F6 r; r.a0 = f2_sub(a.a0, b.a0); r.a1 = f2_sub(a.a1, b.a1); r.a2 = f2_sub(a.a2, b.a2); return r; }
inline F6 f6_neg(const F6& a) { F6 r; r.a0 = f2_neg(a.a0); r.a1 = f2_neg(a.a1); r.a2 = f2_neg(a.a2); return r; }
inline F6 f6_mul_v(const F6& a) { F6 r; r.a0 = f2_mul_xi(a.a2); r.a1 = a.a0; r.a2 = a.a1; return r; }
inline F6 f6_mul(const F6& a, const F6& b) {
  const Fq2 t0 = f2_mul(a.a0, b.a0), t1 = f2_mul(a.a1, b.a1), t2 = f2_mul(a.a2, b.a2);
  F6 r;
  r.a0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.a1, a.a2), f2_add(b.a1, b.a2)), t1), t2)));
  r.a1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a1), f2_add(b.a0, b.a1)), t0), t1), f2_mul_xi(t2));
  r.a2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a2), f2_add(b.a0, b.a2)), t0), t2), t1);
  return r;
}
inline F6 f6_inv(const F6& a) {
  const Fq2 c0 = f2_sub(f2_sqr(a.a0), f2_mul_xi(f2_mul(a.a1, a.a2)));
  const Fq2 c1 = f2_sub(f2_mul_xi(f2_sqr(a.a2)), f2_mul(a.a0, a.a1));
  const Fq2 c2 = f2_sub(f2_sqr(a.a1), f2_mul(a.a0, a.a2));
  const Fq2 t = f2_inv(f2_add(f2_mul(a.a0, c0), f2_mul_xi(f2_add(f2_mul(a.a2, c1), f2_mul(a.a1, c2)))));
  F6 r;
  r.a0 = f2_mul(c0, t); r.a1 = f2_mul(c1, t); r.a2 = f2_mul(c2, t);
  return r;
}

// Fq12 = Fq6[w] / (w^2 - v): c0 + c1 w; as a polynomial in w over Fq2 the coefficient of w^i is (c0.a0, c1.a0, c0.a1, c1.a1, c0.a2, c1.a2)[i]
struct F12 {
  F6 c0, c1;
  static F12 one() { F12 r; r.c0 = F6::one(); r.c1 = F6::zero(); return r; }
  bool operator==(const F12& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool is_one() const { return *this == one(); }
};
inline F12 f12_mul(const F12& a, const F12& b) {
  const F6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
  F12 r;
  r.c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
  r.c0 = f6_add(t0, f6_mul_v(t1));
  return r;
}
inline F12 f12_sqr(const F12& a) {
  const F6 ab = f6_mul(a.c0, a.c1);
  F12 r;
  r.c0 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_v(a.c1))), ab), f6_mul_v(ab));
  r.c1 = f6_add(ab, ab);
  return r;
}
inline F12 f12_conj(const F12& a) { F12 r; r.c0 = a.c0; r.c1 = f6_neg(a.c1); return r; }       // a^(q^6)
inline F12 f12_inv(const F12& a) {
  const F6 t = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1))));
  F12 r;
  r.c0 = f6_mul(a.c0, t);
  r.c1 = f6_neg(f6_mul(a.c1, t));
  return r;
}

// a * (l0 + l2 w^2 + l3 w^3): the shape of every line below.  With L0 = (l0, l2, 0) and L1 = (0, l3, 0) in Fq6 the three Fq6
// products of the Karatsuba step have one or two zero coefficients: 13 instead of 18 Fq2 products.
inline F6 f6_mul_01(const F6& a, const Fq2& b0, const Fq2& b1) {       // a * (b0 + b1 v)
  const Fq2 t0 = f2_mul(a.a0, b0), t1 = f2_mul(a.a1, b1);
  F6 r;
  r.a0 = f2_add(t0, f2_mul_xi(f2_mul(a.a2, b1)));
  r.a1 = f2_sub(f2_sub(f2_mul(f2_add(a.a0, a.a1), f2_add(b0, b1)), t0), t1);
  r.a2 = f2_add(f2_mul(a.a2, b0), t1);
  return r;
}
inline F6 f6_mul_1(const F6& a, const Fq2& b1) {                       // a * (b1 v)
  F6 r;
  r.a0 = f2_mul_xi(f2_mul(a.a2, b1));
  r.a1 = f2_mul(a.a0, b1);
  r.a2 = f2_mul(a.a1, b1);
  return r;
}
inline F12 f12_mul_line(const F12& a, const Fq2& l0, const Fq2& l2, const Fq2& l3) {
  const F6 t0 = f6_mul_01(a.c0, l0, l2), t1 = f6_mul_1(a.c1, l3);
  F12 r;
  r.c1 = f6_sub(f6_sub(f6_mul_01(f6_add(a.c0, a.c1), l0, f2_add(l2, l3)), t0), t1);
  r.c0 = f6_add(t0, f6_mul_v(t1));
  return r;
}

// gamma[i] = xi^(i (q - 1) / 6): (alpha w^i)^q = conj(alpha) gamma[i] w^i
struct FrobeniusConstants {
  Fq2 gamma[6];
  FrobeniusConstants() {
    constexpr uint32_t q[12] = FQ_P;
    uint32_t e[12];
    uint64_t rem = 0;
    for (int i = 11; i >= 0; i--) {                         // (q - 1) / 6; q - 1 only clears bit 0 of the odd q
      const uint64_t cur = (rem << 32) | (i == 0 ? q[0] - 1 : q[i]);
      e[i] = (uint32_t)(cur / 6);
      rem = cur % 6;
    }
    Fq2 xi; xi.c0 = Fq::one(); xi.c1 = Fq::one();
    Fq2 g = Fq2::one();
    for (int i = 12 * 32 - 1; i >= 0; i--) {
      g = f2_sqr(g);
      if ((e[i >> 5] >> (i & 31)) & 1) g = f2_mul(g, xi);
    }
    gamma[0] = Fq2::one();
    for (int i = 1; i < 6; i++) gamma[i] = f2_mul(gamma[i - 1], g);
  }
};
inline const FrobeniusConstants& frobenius_constants() { static const FrobeniusConstants c; return c; }
inline F12 f12_frobenius(const F12& a) {
  const FrobeniusConstants& k = frobenius_constants();
  F12 r;
  r.c0.a0 = f2_conj(a.c0.a0);
  r.c1.a0 = f2_mul(f2_conj(a.c1.a0), k.gamma[1]);
  r.c0.a1 = f2_mul(f2_conj(a.c0.a1), k.gamma[2]);
  r.c1.a1 = f2_mul(f2_conj(a.c1.a1), k.gamma[3]);
  r.c0.a2 = f2_mul(f2_conj(a.c0.a2), k.gamma[4]);
  r.c1.a2 = f2_mul(f2_conj(a.c1.a2), k.gamma[5]);
  return r;
}

constexpr uint64_t ATE_LOOP = 0xd201000000010000ull;        // |x|; the curve parameter x is negative

// f_{|x|, Q}(P) up to factors in proper subfields; P in G1, Q on the twist (both affine, neither checked here)
inline F12 miller_loop(const G1Affine& p, const G2Affine& q) {
  if (p.is_inf() || q.is_inf()) return F12::one();
  G2Jac R; R.x = q.x; R.y = q.y; R.z = Fq2::one();
  F12 f = F12::one();
  for (int i = 62; i >= 0; i--) {
    {
      // tangent at R = (X, Y, Z): (2 Y^2 - 3 X^3) + 3 X^2 Z^2 xP w^2 - 2 Y Z^3 yP w^3
      const Fq2 X2 = f2_sqr(R.x), Z2 = f2_sqr(R.z), Y2 = f2_sqr(R.y);
      const Fq2 X2_3 = f2_add(f2_dbl(X2), X2);
      const Fq2 l0 = f2_sub(f2_dbl(Y2), f2_mul(X2_3, R.x));
      const Fq2 l2 = f2_mul_fq(f2_mul(X2_3, Z2), p.x);
      const Fq2 l3 = f2_neg(f2_mul_fq(f2_dbl(f2_mul(f2_mul(R.y, R.z), Z2)), p.y));
      f = f12_mul_line(f12_sqr(f), l0, l2, l3);
      R = g2_dbl(R);
    }
    if ((ATE_LOOP >> i) & 1) {
      // chord through R = (X, Y, Z) and Q = (x2, y2): H = x2 Z^2 - X, N = y2 Z^3 - Y:
      //   (y2 Z H - N x2) + N xP w^2 - Z H yP w^3
      const Fq2 Z2 = f2_sqr(R.z);
      const Fq2 H = f2_sub(f2_mul(q.x, Z2), R.x);
      const Fq2 N = f2_sub(f2_mul(q.y, f2_mul(Z2, R.z)), R.y);
      if (H.is_zero()) {
        // R = +-Q: cannot happen for a point of order r inside the loop; a caller that hands over anything else gets the
        // degenerate value 1 (an equation between such values proves nothing, and none is accepted: points are subgroup-checked)
        return F12::one();
      }
      const Fq2 ZH = f2_mul(R.z, H);
      const Fq2 l0 = f2_sub(f2_mul(q.y, ZH), f2_mul(N, q.x));
      const Fq2 l2 = f2_mul_fq(N, p.x);
      const Fq2 l3 = f2_neg(f2_mul_fq(ZH, p.y));
      f = f12_mul_line(f, l0, l2, l3);
      R = g2_add_mixed(R, q);
    }
  }
  return f;
}

// g^|x| for g in the cyclotomic subgroup (plain squarings: the verifier does ~320 of them per check)
inline F12 f12_pow_absx(const F12& g) {
  F12 acc = g;
  for (int i = 62; i >= 0; i--) {
    acc = f12_sqr(acc);
    if ((ATE_LOOP >> i) & 1) acc = f12_mul(acc, g);
  }
  return acc;
}

// f^(3 (q^12 - 1) / r)
inline F12 final_exponentiation(const F12& f) {
  // easy part: (q^6 - 1)(q^2 + 1); afterwards the inverse is the conjugate
  F12 g = f12_mul(f12_conj(f), f12_inv(f));
  g = f12_mul(f12_frobenius(f12_frobenius(g)), g);
  // hard part: (x - 1)^2 (x + q)(x^2 + q^2 - 1) + 3 with x = -|x|
  auto pow_x = [](const F12& h) { return f12_conj(f12_pow_absx(h)); };                     // h^x
  auto pow_xm1 = [](const F12& h) { return f12_conj(f12_mul(f12_pow_absx(h), h)); };       // h^(x-1) = 1 / h^(|x|+1)
  const F12 a = pow_xm1(pow_xm1(g));
  const F12 b = f12_mul(pow_x(a), f12_frobenius(a));
  const F12 c = f12_mul(f12_mul(pow_x(pow_x(b)), f12_frobenius(f12_frobenius(b))), f12_conj(b));
  return f12_mul(c, f12_mul(f12_sqr(g), g));
}

}  // namespace pairing
}  // namespace sonic
