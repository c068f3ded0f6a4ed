// Test infrastructure: the plain pairing the verifier used in rounds 1-2, kept as the cross-check of the tower-field pairing in
// sonic_amd/csrc/pairing.hpp (tests/pairing_selftest.cpp).  Affine chord-and-tangent arithmetic on the curve untwisted into
// Fq12 = Fq[w]/(w^12 - 2 w^6 + 2) in the polynomial basis (u = w^6 - 1), one inversion in Fq12 per line, and the exponent
// (q^12 - 1)/r computed with schoolbook integers and applied bit by bit: ~0.17 s per three-pairing check.  It was itself
// cross-checked against oracle/pairing.py (python integers) through the accept / reject tests of tests/test_gpu_parity.py.
#pragma once
#include <vector>
#include "g2.hpp"
#include "constants.hpp"

namespace sonic {
namespace plain {

// ---- Fq12 in the polynomial basis 1, w, ..., w^11 --------------------------------------------------
struct F12 { Fq c[12]; };
F12 f12_zero() { F12 r; for (auto& x : r.c) x = Fq::zero(); return r; }
F12 f12_one() { F12 r = f12_zero(); r.c[0] = Fq::one(); return r; }
bool f12_eq(const F12& a, const F12& b) { for (int i = 0; i < 12; i++) if (a.c[i] != b.c[i]) return false; return true; }
F12 f12_add(const F12& a, const F12& b) { F12 r; for (int i = 0; i < 12; i++) r.c[i] = fp_add(a.c[i], b.c[i]); return r; }
F12 f12_sub(const F12& a, const F12& b) { F12 r; for (int i = 0; i < 12; i++) r.c[i] = fp_sub(a.c[i], b.c[i]); return r; }
F12 f12_small(const F12& a, int k) { F12 r = a; for (int j = 1; j < k; j++) r = f12_add(r, a); return r; }
F12 f12_mul(const F12& a, const F12& b) {
  Fq t[23];
  for (auto& x : t) x = Fq::zero();
  for (int i = 0; i < 12; i++) {
    if (a.c[i].is_zero()) continue;
    for (int j = 0; j < 12; j++) t[i + j] = fp_add(t[i + j], fp_mul(a.c[i], b.c[j]));
  }
  for (int k = 22; k >= 12; k--) {                 // w^12 = 2 w^6 - 2
    Fq c2 = fp_dbl(t[k]);
    t[k - 6] = fp_add(t[k - 6], c2);
    t[k - 12] = fp_sub(t[k - 12], c2);
  }
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = t[i];
  return r;
}
// inverse by the extended Euclidean algorithm on polynomials over Fq
F12 f12_inv(const F12& a) {
  auto deg = [](const std::vector<Fq>& p) { int d = (int)p.size() - 1; while (d > 0 && p[d].is_zero()) d--; return d; };
  std::vector<Fq> lm(13, Fq::zero()), hm(13, Fq::zero()), low(13, Fq::zero()), high(13, Fq::zero());
  lm[0] = Fq::one();
  for (int i = 0; i < 12; i++) low[i] = a.c[i];
  high[0] = fp_dbl(Fq::one()); high[6] = fp_neg(fp_dbl(Fq::one())); high[12] = Fq::one();
  while (deg(low) > 0) {
    // r = high div low
    std::vector<Fq> rem = high, quo(13, Fq::zero());
    const int dl = deg(low);
    const Fq il = fp_inv(low[dl]);
    for (int i = deg(rem) - dl; i >= 0; i--) {
      Fq c = fp_mul(rem[dl + i], il);
      quo[i] = c;
      for (int j = 0; j <= dl; j++) rem[i + j] = fp_sub(rem[i + j], fp_mul(c, low[j]));
    }
    std::vector<Fq> nm = hm, nw = high;
    for (int i = 0; i < 13; i++)
      for (int j = 0; j < 13 - i; j++) {
        nm[i + j] = fp_sub(nm[i + j], fp_mul(lm[i], quo[j]));
        nw[i + j] = fp_sub(nw[i + j], fp_mul(low[i], quo[j]));
      }
    hm = lm; high = low; lm = nm; low = nw;
  }
  const Fq c = fp_inv(low[0]);
  F12 r;
  for (int i = 0; i < 12; i++) r.c[i] = fp_mul(lm[i], c);
  return r;
}
// c0 + c1 u with u = w^6 - 1
F12 f12_from_f2(const Fq2& a) { F12 r = f12_zero(); r.c[0] = fp_sub(a.c0, a.c1); r.c[6] = a.c1; return r; }
F12 f12_from_fq(const Fq& a) { F12 r = f12_zero(); r.c[0] = a; return r; }

struct E12 { F12 x, y; };

F12 line(const E12& p1, const E12& p2, const E12& t) {
  F12 m;
  if (!f12_eq(p1.x, p2.x)) m = f12_mul(f12_sub(p2.y, p1.y), f12_inv(f12_sub(p2.x, p1.x)));
  else if (f12_eq(p1.y, p2.y)) m = f12_mul(f12_small(f12_mul(p1.x, p1.x), 3), f12_inv(f12_small(p1.y, 2)));
  else return f12_sub(t.x, p1.x);
  return f12_sub(f12_mul(m, f12_sub(t.x, p1.x)), f12_sub(t.y, p1.y));
}
E12 e12_add(const E12& p1, const E12& p2) {
  F12 m;
  if (f12_eq(p1.x, p2.x) && f12_eq(p1.y, p2.y)) m = f12_mul(f12_small(f12_mul(p1.x, p1.x), 3), f12_inv(f12_small(p1.y, 2)));
  else m = f12_mul(f12_sub(p2.y, p1.y), f12_inv(f12_sub(p2.x, p1.x)));
  E12 r;
  r.x = f12_sub(f12_sub(f12_mul(m, m), p1.x), p2.x);
  r.y = f12_sub(f12_mul(m, f12_sub(p1.x, r.x)), p1.y);
  return r;
}

F12 miller_loop(const G1Affine& p, const G2Affine& q) {
  if (p.is_inf() || q.is_inf()) return f12_one();
  F12 w2 = f12_zero(), w3 = f12_zero();
  w2.c[2] = Fq::one(); w3.c[3] = Fq::one();
  static const F12 w2i = f12_inv(w2), w3i = f12_inv(w3);
  E12 Q, P, R;
  Q.x = f12_mul(f12_from_f2(q.x), w2i);
  Q.y = f12_mul(f12_from_f2(q.y), w3i);
  P.x = f12_from_fq(p.x); P.y = f12_from_fq(p.y);
  R = Q;
  F12 f = f12_one();
  const uint64_t loop = 0xd201000000010000ull;
  for (int i = 62; i >= 0; i--) {
    f = f12_mul(f12_mul(f, f), line(R, R, P));
    R = e12_add(R, R);
    if ((loop >> i) & 1) { f = f12_mul(f, line(R, Q, P)); R = e12_add(R, Q); }
  }
  return f;
}

// f^((q^12 - 1) / r): the exponent is computed once with schoolbook big-integer arithmetic
const std::vector<uint32_t>& final_exponent() {
  static std::vector<uint32_t> e;
  if (!e.empty()) return e;
  constexpr uint32_t q[12] = FQ_P, r[8] = FR_P;
  std::vector<uint32_t> num(1, 1);
  for (int k = 0; k < 12; k++) {                        // num = q^12
    std::vector<uint32_t> t(num.size() + 12, 0);
    for (size_t i = 0; i < num.size(); i++) {
      uint64_t c = 0;
      for (int j = 0; j < 12; j++) { c += (uint64_t)num[i] * q[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
      for (size_t j = i + 12; c; j++) { c += t[j]; t[j] = (uint32_t)c; c >>= 32; }
    }
    num = t;
  }
  num[0] -= 1;                                          // q^12 is odd, so no borrow
  // long division by r, bit by bit (4600 x 8 limb steps)
  std::vector<uint32_t> quo(num.size(), 0);
  uint32_t rem[9] = {0};
  for (long b = (long)num.size() * 32 - 1; b >= 0; b--) {
    for (int i = 8; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
    rem[0] = (rem[0] << 1) | ((num[b >> 5] >> (b & 31)) & 1);
    bool ge = rem[8] != 0;
    if (!ge) { ge = true; for (int i = 7; i >= 0; i--) { if (rem[i] != r[i]) { ge = rem[i] > r[i]; break; } } }
    if (ge) {
      uint64_t br = 0;
      for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)rem[i] - r[i] - br; rem[i] = (uint32_t)d; br = (d >> 32) & 1; }
      rem[8] -= (uint32_t)br;
      quo[b >> 5] |= 1u << (b & 31);
    }
  }
  while (quo.size() > 1 && quo.back() == 0) quo.pop_back();
  e = quo;
  return e;
}
F12 final_exp(const F12& f) {
  const auto& e = final_exponent();
  F12 acc = f12_one();
  bool started = false;
  for (long i = (long)e.size() * 32 - 1; i >= 0; i--) {
    if (started) acc = f12_mul(acc, acc);
    if ((e[i >> 5] >> (i & 31)) & 1) { acc = f12_mul(acc, f); started = true; }
  }
  return acc;
}

}  // namespace plain
}  // namespace sonic
