// BLS12-381 Fq (381-bit, 12 x u32) and Fr (255-bit, 8 x u32) in Montgomery form for gfx950.
//
// Replaces the arithmetic the reference inherits from galois-field-1.0.1 (`Prime p` over Natural)
// as exercised at src/Sonic/CommitmentScheme.hs:26-29,43-48, src/Sonic/Utils.hs:18,21 and
// src/Sonic/Constraints.hs:28-65.  CDNA4 has no 64x64 multiplier: the native wide multiply is
// v_mad_u64_u32 (32x32+64 -> 64), so limbs are 32-bit and every inner step is one such MAD.
// The modulus limbs are compile-time literals (no VGPRs spent on them).
//
// The same source compiles for the host (g++, tests/host/) so the limb logic is unit-tested on CPU.
//
// LAZY RANGE (Fq, device, assembly build): 4q < 2^384, so the Montgomery product of two values < 2q is again < 2q
// without the final conditional subtraction ("almost Montgomery").  On the device Fq values therefore live in [0, 2q):
// the product drops 30 instructions, add / sub correct by 2q, and 0 has the two representations 0 and q, which is_zero()
// and == accept.  Canonical (< q) values are a special case, so canonical inputs need no conversion; what leaves the
// device for the host's portable code is brought back below q (fp_canonical: slots, from_mont).  Fr (4r > 2^256) and
// everything on the host stay canonical.
#pragma once
#include <stdint.h>
#include "constants.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(SONIC_NO_ASM_MUL)
#define SONIC_FQ_LAZY 1
#else
#define SONIC_FQ_LAZY 0
#endif

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
// The Montgomery product is the one function kept out of line on the device: a fully inlined
// point addition is > 100 KB of code (14 products x ~0.9 K instructions), far beyond the 64 KB
// instruction cache; as a call it takes 24 VGPR arguments and returns 12.
#define HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define HD inline __attribute__((always_inline))
#define HD_NOINLINE inline
#endif

namespace sonic {

struct FqParams {
  static constexpr int N = FQ_LIMBS;
  static constexpr uint32_t INV = FQ_INV;
  static constexpr bool LAZY = SONIC_FQ_LAZY != 0;
  static HD constexpr uint32_t p(int i) { constexpr uint32_t v[N] = FQ_P; return v[i]; }
  static HD constexpr uint32_t p2(int i) { constexpr uint32_t v[N] = FQ_P2; return v[i]; }
  static HD constexpr uint32_t one(int i) { constexpr uint32_t v[N] = FQ_ONE; return v[i]; }
  static HD constexpr uint32_t r2(int i) { constexpr uint32_t v[N] = FQ_R2; return v[i]; }
};
struct FrParams {
  static constexpr int N = FR_LIMBS;
  static constexpr uint32_t INV = FR_INV;
  static constexpr bool LAZY = false;
  static HD constexpr uint32_t p(int i) { constexpr uint32_t v[N] = FR_P; return v[i]; }
  static HD constexpr uint32_t p2(int i) { constexpr uint32_t v[N] = FR_P2; return v[i]; }
  static HD constexpr uint32_t one(int i) { constexpr uint32_t v[N] = FR_ONE; return v[i]; }
  static HD constexpr uint32_t r2(int i) { constexpr uint32_t v[N] = FR_R2; return v[i]; }
};

template <class P>
struct Fp {
  static constexpr int N = P::N;
  uint32_t l[N];

  static HD Fp zero() { Fp r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
  static HD Fp one() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::one(i); return r; }
  static HD Fp r2() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::r2(i); return r; }
  static HD Fp modulus() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::p(i); return r; }

  // congruent to 0: all limbs 0, or (lazy range) equal to the modulus
  HD bool is_zero() const {
    uint32_t t = 0;
    for (int i = 0; i < N; i++) t |= l[i];
    if constexpr (P::LAZY) {
      uint32_t u = 0;
      for (int i = 0; i < N; i++) u |= l[i] ^ P::p(i);
      return t == 0 || u == 0;
    }
    return t == 0;
  }
  // all limbs zero: enough wherever zero only ever appears as the literal 0 (the infinity encodings)
  HD bool is_zero_strict() const { uint32_t t = 0; for (int i = 0; i < N; i++) t |= l[i]; return t == 0; }
  HD bool operator==(const Fp& o) const;
  HD bool operator!=(const Fp& o) const { return !(*this == o); }
};

// r = a - p if a >= p (a < 2p), branch-free select
template <class P>
HD void fp_reduce_once(Fp<P>& a) {
  constexpr int N = P::N;
  uint32_t t[N];
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - P::p(i) - br;
    t[i] = (uint32_t)d;
    br = (d >> 32) & 1;
  }
  if (!br) {
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = t[i];
  }
}

template <class P>
HD Fp<P> fp_add_generic(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  Fp<P> r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)a.l[i] + b.l[i];
    r.l[i] = (uint32_t)c;
    c >>= 32;
  }
  // both moduli leave >= 1 spare bit in N limbs, so a + b < 2p < 2^(32N): no carry out
  fp_reduce_once(r);
  return r;
}

template <class P>
HD Fp<P> fp_sub_generic(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N;
  Fp<P> r;
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
    r.l[i] = (uint32_t)d;
    br = (d >> 32) & 1;
  }
  uint32_t mask = (uint32_t)0 - (uint32_t)br;  // add p back when a < b
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    c += (uint64_t)r.l[i] + (P::p(i) & mask);
    r.l[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

template <class P>
HD Fp<P> fp_neg(const Fp<P>& a) {
  constexpr int N = P::N;
  Fp<P> r;
  uint64_t br = 0;
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < N; i++) nz |= a.l[i];
  uint32_t mask = nz ? 0xffffffffu : 0u;
#pragma unroll
  for (int i = 0; i < N; i++) {
    // lazy range: 2p - a stays in [0, 2p) for 0 < a < 2p
    uint64_t d = (uint64_t)((P::LAZY ? P::p2(i) : P::p(i)) & mask) - a.l[i] - br;
    r.l[i] = (uint32_t)d;
    br = (d >> 32) & 1;
  }
  return r;
}

// Montgomery product a*b*R^-1 mod p, operand-scanning CIOS over 32-bit limbs.
// Each inner step is one 32x32+32+32 -> 64 (fits: (2^32-1)^2 + 2(2^32-1) = 2^64-1).
template <class P>
HD_NOINLINE Fp<P> fp_mul_generic(const Fp<P> a, const Fp<P> b) {
  constexpr int N = P::N;
  uint32_t t[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.l[i];
#pragma unroll
    for (int j = 0; j < N; j++) {
      c = (uint64_t)a.l[j] * bi + t[j] + c;
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[N];
    t[N] = (uint32_t)c;
    t[N + 1] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * P::INV;
    c = ((uint64_t)m * P::p(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < N; j++) {
      c = (uint64_t)m * P::p(j) + t[j] + c;
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[N];
    t[N - 1] = (uint32_t)c;
    t[N] = t[N + 1] + (uint32_t)(c >> 32);
  }
  // p < 2^(32N-1) so the CIOS result is < 2p and t[N] == 0
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < N; i++) r.l[i] = t[i];
  fp_reduce_once(r);
  return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// The same product for the host (the verifier's pairings, the tails of the MSMs): CIOS over N/2 64-bit limbs with 128-bit
// intermediates, ~6x the speed of the 32-bit loop on a CPU.  Same representation (little-endian limbs), same R, result below p.
template <class P>
inline Fp<P> fp_mul_host64(const Fp<P>& a, const Fp<P>& b) {
  constexpr int N = P::N, M = N / 2;
  static_assert(N % 2 == 0, "even limb count");
  typedef unsigned __int128 u128;
  constexpr uint64_t p0 = (uint64_t)P::p(0) | ((uint64_t)P::p(1) << 32);
  constexpr uint64_t x32 = (uint64_t)(uint32_t)(0u - P::INV);                 // p^-1 mod 2^32
  constexpr uint64_t inv = 0 - x32 * (2 - p0 * x32);                          // -p^-1 mod 2^64 (one Newton step)
  uint64_t A[M], B[M], Pm[M], t[M + 2];
  for (int i = 0; i < M; i++) {
    A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
    B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
    Pm[i] = (uint64_t)P::p(2 * i) | ((uint64_t)P::p(2 * i + 1) << 32);
  }
  for (int i = 0; i < M + 2; i++) t[i] = 0;
  for (int i = 0; i < M; i++) {
    u128 c = 0;
    for (int j = 0; j < M; j++) {
      c += (u128)A[j] * B[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[M];
    t[M] = (uint64_t)c;
    t[M + 1] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * inv;
    c = ((u128)m * Pm[0] + t[0]) >> 64;
    for (int j = 1; j < M; j++) {
      c += (u128)m * Pm[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[M];
    t[M - 1] = (uint64_t)c;
    t[M] = t[M + 1] + (uint64_t)(c >> 64);
  }
  Fp<P> r;
  for (int i = 0; i < M; i++) { r.l[2 * i] = (uint32_t)t[i]; r.l[2 * i + 1] = (uint32_t)(t[i] >> 32); }
  fp_reduce_once(r);
  return r;
}
#endif

}  // namespace sonic
#include "mont_asm.hpp"   // device only: hand-scheduled gfx950 routines (tools/gen_mont_asm.py)
namespace sonic {

// On the device the product is the generated assembly routine (private calling convention, no
// stack, 288 MADs + ~390 other VALU for Fq); on the host, and with -DSONIC_NO_ASM_MUL, the C++ loop.
template <class P>
HD Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SONIC_NO_ASM_MUL)
  if constexpr (P::N == FQ_LIMBS) return sonic_mont_mul_fq_call(a, b);
  else return sonic_mont_mul_fr_call(a, b);
#elif defined(__HIP_DEVICE_COMPILE__)
  return fp_mul_generic(a, b);
#else
  return fp_mul_host64(a, b);
#endif
}

// add / sub on the device: two interleaved carry chains in one asm block (Fq: 48 VALU instead of the ~125 the compiler emits for
// the portable loops; Fr, round 4: 32, canonical range -- the NTT butterflies and the scans are made of these); the host keeps
// the portable code.
template <class P>
HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SONIC_NO_ASM_MUL)
  if constexpr (P::N == FQ_LIMBS) return sonic_fq_add_asm<P>(a, b);
  else return sonic_fr_add_asm<P>(a, b);
#else
  return fp_add_generic(a, b);
#endif
}
template <class P>
HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SONIC_NO_ASM_MUL)
  if constexpr (P::N == FQ_LIMBS) return sonic_fq_sub_asm<P>(a, b);
  else return sonic_fr_sub_asm<P>(a, b);
#else
  return fp_sub_generic(a, b);
#endif
}
template <class P>
HD Fp<P> fp_dbl(const Fp<P>& a) { return fp_add(a, a); }

// a^2: on the device Fq has its own routine (78 instead of 144 partial products, mont_asm.hpp)
template <class P>
HD Fp<P> fp_sqr(const Fp<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SONIC_NO_ASM_MUL)
  if constexpr (P::N == FQ_LIMBS) return sonic_mont_sqr_fq_call(a);
  else return fp_mul(a, a);
#else
  return fp_mul(a, a);
#endif
}

template <class P>
HD Fp<P> fp_to_mont(const Fp<P>& a) { return fp_mul(a, Fp<P>::r2()); }

// representative below the modulus (identity outside the lazy range)
template <class P>
HD Fp<P> fp_canonical(const Fp<P>& a) {
  Fp<P> r = a;
  if constexpr (P::LAZY) fp_reduce_once(r);
  return r;
}

template <class P>
HD bool Fp<P>::operator==(const Fp<P>& o) const {
  if constexpr (P::LAZY) return fp_sub(*this, o).is_zero();
  uint32_t t = 0;
  for (int i = 0; i < N; i++) t |= l[i] ^ o.l[i];
  return t == 0;
}

// standard form, canonical: (a + m p) / R <= p for a < 2p, and = p only for a in {0, p}
template <class P>
HD Fp<P> fp_from_mont(const Fp<P>& a) {
  Fp<P> o = Fp<P>::zero();
  o.l[0] = 1;
  return fp_canonical(fp_mul(a, o));
}

// a^e, e a small non-negative integer (square-and-multiply, MSB first)
template <class P>
HD Fp<P> fp_pow_u64(const Fp<P>& a, uint64_t e) {
  Fp<P> acc = Fp<P>::one();
  bool started = false;
  for (int i = 63; i >= 0; i--) {
    if (started) acc = fp_sqr(acc);
    if ((e >> i) & 1) { acc = fp_mul(acc, a); started = true; }
  }
  return acc;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host only: a^-1 by the binary extended Euclidean algorithm (odd modulus; variable time, which is fine: every value the host
// inverts is public -- the normalisation of a proof element).  ~2 x bits shift / subtract rounds on N limbs: ~15 us for Fq
// against ~170 us for the Fermat power with the portable product.  Montgomery in, Montgomery out: the algorithm inverts the
// stored integer a R, giving a^-1 R^-1; two products with R^2 bring that to a^-1 R.
template <class P>
inline Fp<P> fp_inv_euclid(const Fp<P>& a) {
  constexpr int N = P::N;
  if (a.is_zero()) return Fp<P>::zero();
  uint32_t u[N], v[N], x1[N], x2[N], pm[N];
  for (int i = 0; i < N; i++) { u[i] = a.l[i]; v[i] = pm[i] = P::p(i); x1[i] = 0; x2[i] = 0; }
  x1[0] = 1;
  auto shr1 = [&](uint32_t* w) { for (int i = 0; i < N - 1; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31); w[N - 1] >>= 1; };
  auto add = [&](uint32_t* w, const uint32_t* y) { uint64_t c = 0; for (int i = 0; i < N; i++) { c += (uint64_t)w[i] + y[i]; w[i] = (uint32_t)c; c >>= 32; } };
  auto sub = [&](uint32_t* w, const uint32_t* y) { uint64_t br = 0; for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)w[i] - y[i] - br; w[i] = (uint32_t)d; br = (d >> 32) & 1; } return br; };
  auto geq = [&](const uint32_t* w, const uint32_t* y) { for (int i = N - 1; i >= 0; i--) { if (w[i] != y[i]) return w[i] > y[i]; } return true; };
  auto is_one = [&](const uint32_t* w) { uint32_t t = w[0] ^ 1u; for (int i = 1; i < N; i++) t |= w[i]; return t == 0; };
  auto halve_mod = [&](uint32_t* w) { if (w[0] & 1u) add(w, pm); shr1(w); };      // both moduli leave a spare bit: w + p < 2^(32N)
  auto sub_mod = [&](uint32_t* w, const uint32_t* y) { if (sub(w, y)) add(w, pm); };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1u)) { shr1(u); halve_mod(x1); }
    while (!(v[0] & 1u)) { shr1(v); halve_mod(x2); }
    if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); } else { sub(v, u); sub_mod(x2, x1); }
  }
  Fp<P> r;
  const uint32_t* res = is_one(u) ? x1 : x2;
  for (int i = 0; i < N; i++) r.l[i] = res[i];
  return fp_mul(fp_mul(r, Fp<P>::r2()), Fp<P>::r2());
}
#endif

// a^-1: on the device a^(p-2) (Fermat, ~1.5 * bits multiplications; used once per MSM / per batch), on the host the Euclidean
// routine above.  0^-1 := 0 on both sides.
template <class P>
HD Fp<P> fp_inv(const Fp<P>& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return fp_inv_euclid(a);
#else
  constexpr int N = P::N;
  uint32_t e[N];
  uint64_t br = 2;
  for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)P::p(i) - br;
    e[i] = (uint32_t)d;
    br = (d >> 32) & 1;
  }
  Fp<P> acc = Fp<P>::one();
  bool started = false;
  for (int i = 32 * N - 1; i >= 0; i--) {
    if (started) acc = fp_sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) { acc = fp_mul(acc, a); started = true; }
  }
  return acc;
#endif
}

// canonical check: a < p (standard form)
template <class P>
HD bool fp_is_canonical(const Fp<P>& a) {
  constexpr int N = P::N;
  for (int i = N - 1; i >= 0; i--) {
    if (a.l[i] < P::p(i)) return true;
    if (a.l[i] > P::p(i)) return false;
  }
  return false;
}

typedef Fp<FqParams> Fq;
typedef Fp<FrParams> Fr;

}  // namespace sonic
