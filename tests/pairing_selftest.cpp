// Host self-test of the verifier's pairing (sonic_amd/csrc/pairing.hpp), built with g++ by tests/test_pairing_host.py:
//   * the host's 64-bit-limb Montgomery product (field.hpp, fp_mul_host64) against the portable 32-bit loop, Fq and Fr;
//   * the host inversion and the shared-inversion normalisations of G1 results (g1_host.hpp) against the one-at-a-time ones;
//   * the tower arithmetic against itself (inverse, squaring vs product, sparse line product vs full product, Frobenius^12 = id,
//     Frobenius = q-th power on a random element);
//   * value: final_exponentiation(miller_loop(P, Q)) == plain pairing(P, Q)^3 for random multiples P = a G1, Q = b G2
//     (tests/pairing_plain.hpp: the polynomial-basis pairing the verifier used before, exponent (q^12 - 1)/r);
//   * bilinearity and non-degeneracy: e(aP, bQ) == e(P, Q)^(ab), e(P, Q) != 1, e(P, Q)^r == 1;
//   * a pcV-shaped product: e(W, h^{alpha x}) e(g^v W^{-z}, h^alpha) e(-F, h^{x^k}) == 1 for a polynomial commitment made from a
//     known trapdoor, and != 1 after changing v.
// Prints "pairing selftest ok" and exits 0, or says what failed.
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "pairing.hpp"
#include "g1_host.hpp"
#include "pairing_plain.hpp"

using namespace sonic;
namespace pg = sonic::pairing;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd64() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static Fq rnd_fq() { Fq a; for (int i = 0; i < 12; i++) a.l[i] = (uint32_t)rnd64(); a.l[11] &= 0x0fffffffu; return fp_to_mont(a); }
static Fq2 rnd_f2() { Fq2 a; a.c0 = rnd_fq(); a.c1 = rnd_fq(); return a; }
static pg::F6 rnd_f6() { pg::F6 a; a.a0 = rnd_f2(); a.a1 = rnd_f2(); a.a2 = rnd_f2(); return a; }
static pg::F12 rnd_f12() { pg::F12 a; a.c0 = rnd_f6(); a.c1 = rnd_f6(); return a; }

#define CHECK(cond, what) do { if (!(cond)) { printf("FAILED: %s\n", what); return 1; } } while (0)

static G1Affine g1_gen() {
  constexpr uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  G1Affine g;
  for (int i = 0; i < 12; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
  return g;
}
// the standard G2 generator (standard form, little-endian limbs; the same constants as oracle/pairing.py and EIP-2537)
static G2Affine g2_gen() {
  static const char* hx0 = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
  static const char* hx1 = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e";
  static const char* hy0 = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
  static const char* hy1 = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";
  auto parse = [](const char* h) {
    Fq a;
    for (int i = 0; i < 12; i++) {
      char buf[9]; for (int j = 0; j < 8; j++) buf[j] = h[(11 - i) * 8 + j]; buf[8] = 0;
      a.l[i] = (uint32_t)strtoul(buf, nullptr, 16);
    }
    return fp_to_mont(a);
  };
  G2Affine g;
  g.x.c0 = parse(hx0); g.x.c1 = parse(hx1); g.y.c0 = parse(hy0); g.y.c1 = parse(hy1);
  return g;
}
static G1Affine g1_mul_u64(const G1Affine& p, uint64_t k) {
  G1XYZZ acc = G1XYZZ::inf();
  for (int i = 63; i >= 0; i--) { acc = g1_dbl(acc); if ((k >> i) & 1) acc = g1_add_mixed(acc, p); }
  return g1_to_affine(acc);
}
static G2Affine g2_mul_u64(const G2Affine& p, uint64_t k) {
  G2Jac acc = G2Jac::inf();
  for (int i = 63; i >= 0; i--) { acc = g2_dbl(acc); if ((k >> i) & 1) acc = g2_add_mixed(acc, p); }
  return g2_to_affine(acc);
}
static pg::F12 f12_pow_limbs(const pg::F12& a, const uint32_t* e, int n) {
  pg::F12 acc = pg::F12::one();
  for (int i = n * 32 - 1; i >= 0; i--) { acc = pg::f12_sqr(acc); if ((e[i >> 5] >> (i & 31)) & 1) acc = pg::f12_mul(acc, a); }
  return acc;
}
// plain (polynomial basis 1, w, .., w^11 with u = w^6 - 1) -> tower: coefficient of w^i is c[i] + c[i+6] (u + 1)
static pg::F12 from_plain(const plain::F12& a) {
  Fq2 al[6];
  for (int i = 0; i < 6; i++) { al[i].c0 = fp_add(a.c[i], a.c[i + 6]); al[i].c1 = a.c[i + 6]; }
  pg::F12 r;
  r.c0.a0 = al[0]; r.c1.a0 = al[1]; r.c0.a1 = al[2]; r.c1.a1 = al[3]; r.c0.a2 = al[4]; r.c1.a2 = al[5];
  return r;
}
static pg::F12 pair(const G1Affine& p, const G2Affine& q) { return pg::final_exponentiation(pg::miller_loop(p, q)); }

int main() {
  // ---- the host's 64-bit-limb Montgomery product against the portable 32-bit loop (field.hpp) ----
  {
    auto rnd_fr = [] { Fr a; for (int i = 0; i < 8; i++) a.l[i] = (uint32_t)rnd64(); a.l[7] &= 0x3fffffffu; return a; };
    for (int it = 0; it < 2000; it++) {
      Fq a, b;
      for (int i = 0; i < 12; i++) { a.l[i] = (uint32_t)rnd64(); b.l[i] = (uint32_t)rnd64(); }
      a.l[11] &= 0x0fffffffu; b.l[11] &= 0x0fffffffu;
      if (it == 0) { a = Fq::zero(); }
      if (it == 1) { constexpr uint32_t q[12] = FQ_P; for (int i = 0; i < 12; i++) a.l[i] = b.l[i] = q[i]; a.l[0] -= 1; b.l[0] -= 1; }   // (q-1)^2
      if (it == 2) { constexpr uint32_t q2[12] = FQ_P2; for (int i = 0; i < 12; i++) a.l[i] = b.l[i] = q2[i]; a.l[0] -= 1; b.l[0] -= 1; }  // lazy-range operands 2q-1
      CHECK(fp_mul_host64(a, b) == fp_mul_generic(a, b), "Fq: 64-bit-limb product == 32-bit-limb product");
      const Fr c = rnd_fr(), d = rnd_fr();
      CHECK(fp_mul_host64(c, d) == fp_mul_generic(c, d), "Fr: 64-bit-limb product == 32-bit-limb product");
    }
  }
  // ---- host inversion (binary Euclid, field.hpp) and the shared-inversion normalisations of g1_host.hpp ----
  {
    for (int it = 0; it < 200; it++) {
      const Fq a = rnd_fq();
      CHECK(fp_mul(a, fp_inv(a)) == Fq::one(), "Fq: a * a^-1 == 1");
      Fr b; for (int i = 0; i < 8; i++) b.l[i] = (uint32_t)rnd64(); b.l[7] &= 0x3fffffffu; b = fp_to_mont(b);
      CHECK(b.is_zero() || fp_mul(b, fp_inv(b)) == Fr::one(), "Fr: a * a^-1 == 1");
    }
    CHECK(fp_inv(Fq::zero()).is_zero() && fp_inv(Fq::one()) == Fq::one(), "0^-1 := 0, 1^-1 == 1");
    std::vector<G1XYZZ> pts;
    G1XYZZ a = g1_dbl_affine(g1_gen());
    for (int i = 0; i < 19; i++) {
      a = g1_add_mixed(g1_dbl(a), g1_gen());
      pts.push_back(i % 7 == 3 ? G1XYZZ::inf() : a);
    }
    pts.push_back(G1XYZZ::from_affine(g1_gen()));
    std::vector<uint8_t> one(96 * pts.size()), all(96 * pts.size());
    std::vector<G1Affine> aff(pts.size());
    for (size_t i = 0; i < pts.size(); i++) g1_canonical_bytes_host(pts[i], &one[96 * i]);
    g1_canonical_bytes_host_batch(pts.data(), (int)pts.size(), all.data());
    CHECK(one == all, "canonical bytes: shared inversion == one inversion per point (with points at infinity)");
    g1_batch_affine_host(pts.data(), (long)pts.size(), aff.data());
    for (size_t i = 0; i < pts.size(); i++) {
      const G1Affine w = g1_to_affine(pts[i]);
      CHECK(aff[i].x == w.x && aff[i].y == w.y, "batch affine == g1_to_affine");
    }
    std::vector<uint8_t> none(1);
    g1_canonical_bytes_host_batch(pts.data(), 0, none.data());       // empty batch: nothing written, no inversion of an empty product gone wrong
  }
  // ---- tower arithmetic ----
  for (int it = 0; it < 4; it++) {
    const pg::F12 a = rnd_f12(), b = rnd_f12(), c = rnd_f12();
    CHECK(pg::f12_mul(a, pg::f12_inv(a)).is_one(), "a * a^-1 == 1");
    CHECK(pg::f12_sqr(a) == pg::f12_mul(a, a), "a^2 == a * a");
    CHECK(pg::f12_mul(pg::f12_mul(a, b), c) == pg::f12_mul(a, pg::f12_mul(b, c)), "associativity");
    CHECK(pg::f12_mul(a, b) == pg::f12_mul(b, a), "commutativity");
    const Fq2 l0 = rnd_f2(), l2 = rnd_f2(), l3 = rnd_f2();
    pg::F12 line; line.c0 = pg::F6::zero(); line.c1 = pg::F6::zero();
    line.c0.a0 = l0; line.c0.a1 = l2; line.c1.a1 = l3;
    CHECK(pg::f12_mul_line(a, l0, l2, l3) == pg::f12_mul(a, line), "sparse line product == full product");
    pg::F12 f = a;
    for (int k = 0; k < 12; k++) f = pg::f12_frobenius(f);
    CHECK(f == a, "Frobenius^12 == id");
    pg::F12 f6 = a;
    for (int k = 0; k < 6; k++) f6 = pg::f12_frobenius(f6);
    CHECK(f6 == pg::f12_conj(a), "Frobenius^6 == conjugation");
    CHECK(pg::f12_frobenius(pg::f12_mul(a, b)) == pg::f12_mul(pg::f12_frobenius(a), pg::f12_frobenius(b)), "Frobenius is multiplicative");
    if (it == 0) {
      constexpr uint32_t q[12] = FQ_P;
      CHECK(pg::f12_frobenius(a) == f12_pow_limbs(a, q, 12), "Frobenius == q-th power");
    }
  }
  // the tower embedding of the plain representation is a ring homomorphism
  {
    plain::F12 a = plain::f12_zero(), b = plain::f12_zero();
    for (int i = 0; i < 12; i++) { a.c[i] = rnd_fq(); b.c[i] = rnd_fq(); }
    CHECK(from_plain(plain::f12_mul(a, b)) == pg::f12_mul(from_plain(a), from_plain(b)), "plain -> tower is multiplicative");
  }
  // ---- pairing values ----
  const G1Affine g1 = g1_gen();
  const G2Affine g2 = g2_gen();
  {
    Fq2 four_xi; four_xi.c0 = fp_dbl(fp_dbl(Fq::one())); four_xi.c1 = four_xi.c0;
    CHECK(f2_sqr(g2.y) == f2_add(f2_mul(f2_sqr(g2.x), g2.x), four_xi), "G2 generator is on the twist");
  }
  const uint64_t a = rnd64() | 1, b = rnd64() | 1;
  const G1Affine P = g1_mul_u64(g1, a);
  const G2Affine Q = g2_mul_u64(g2, b);
  auto t0 = std::chrono::steady_clock::now();
  const pg::F12 e_gg = pair(g1, g2);
  const double ms_fast = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const pg::F12 e_pq = pair(P, Q);
  CHECK(!e_gg.is_one(), "e(g1, g2) != 1");
  {
    constexpr uint32_t r[8] = FR_P;
    CHECK(f12_pow_limbs(e_gg, r, 8).is_one(), "e(g1, g2)^r == 1");
  }
  {
    // e(a g1, b g2) == e(g1, g2)^(a b)
    const unsigned __int128 ab = (unsigned __int128)a * b;
    uint32_t e[4] = {(uint32_t)ab, (uint32_t)(ab >> 32), (uint32_t)(ab >> 64), (uint32_t)(ab >> 96)};
    CHECK(e_pq == f12_pow_limbs(e_gg, e, 4), "bilinearity: e(a g1, b g2) == e(g1, g2)^(a b)");
    CHECK(pair(P, g2) == pair(g1, g2_mul_u64(g2, a)), "e(a g1, g2) == e(g1, a g2)");
  }
  t0 = std::chrono::steady_clock::now();
  const plain::F12 s_gg = plain::final_exp(plain::miller_loop(g1, g2));
  const double ms_plain = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  {
    const pg::F12 s = from_plain(s_gg);
    CHECK(e_gg == pg::f12_mul(pg::f12_sqr(s), s), "value: pairing(g1, g2) == plain pairing(g1, g2)^3");
    const pg::F12 s2 = from_plain(plain::final_exp(plain::miller_loop(P, Q)));
    CHECK(e_pq == pg::f12_mul(pg::f12_sqr(s2), s2), "value: pairing(P, Q) == plain pairing(P, Q)^3");
  }
  // ---- a pcV-shaped product (CommitmentScheme.hs:58-68) from a known trapdoor, in the exponent of g1 / g2 ----
  {
    // f(X) = c X^k with commitment F = g^{alpha c x^{k + d - max}} ... in small numbers: take x, alpha, z, one coefficient; all
    // exponents below 2^64 so that 64-bit multiples suffice: F' = alpha * f(x) * x^shift, W' = (f(x) - f(z)) / (x - z)
    const uint64_t x = 5, alpha = 7, z = 3, c = 11, shift = 2;       // f(X) = c X^2, d - max = shift
    const uint64_t fx = c * x * x, fz = c * z * z;
    const uint64_t Wp = (fx - fz) / (x - z);                          // c (x + z)
    const uint64_t Fp = alpha * fx * 25;                              // alpha f(x) x^shift, x^shift = 25
    (void)shift;
    const G1Affine F = g1_mul_u64(g1, Fp), W = g1_mul_u64(g1, Wp);
    const G2Affine h_alpha = g2_mul_u64(g2, alpha), h_alpha_x = g2_mul_u64(g2, alpha * x);
    // pcV: e(W, h^{alpha x}) e(g^v W^{-z}, h^alpha) == e(F, h^{x^{-d+max}}) with h^{x^{-shift}}; multiply the equation through by
    // x^shift in the exponent of the G2 side instead (no inverses of small numbers needed): use h' = h^{x^shift} on the left
    const G2Affine hl_ax = g2_mul_u64(h_alpha_x, 25), hl_a = g2_mul_u64(h_alpha, 25);
    auto check = [&](uint64_t v) {
      // g^v W^{-z}
      G1XYZZ left = G1XYZZ::from_affine(g1_mul_u64(g1, v));
      left = g1_add_mixed(left, g1_neg(g1_mul_u64(W, z)));
      const G1Affine L = g1_to_affine(left);
      pg::F12 f = pg::f12_mul(pg::f12_mul(pg::miller_loop(W, hl_ax), pg::miller_loop(L, hl_a)), pg::miller_loop(g1_neg(F), g2));
      return pg::final_exponentiation(f).is_one();
    };
    CHECK(check(fz), "pcV-shaped product accepts the true evaluation");
    CHECK(!check(fz + 1), "pcV-shaped product rejects another evaluation");
  }
  printf("pairing selftest ok (one pairing: %.2f ms; plain: %.1f ms)\n", ms_fast, ms_plain);
  return 0;
}
