// The BLS12-381 endomorphism phi(x, y) = (beta x, y) = lambda (x, y) on G1, used to HALVE THE NUMBER OF TABLE WINDOWS of the
// fixed-base MSM where the full window tables do not fit in HBM:
//
//     s P = s1 P + s2 phi(P),   s = s1 + lambda s2,   0 <= s1 < lambda < 2^128,  0 <= s2 <= (r - 1) / lambda < 2^128
//
// and, phi being a group homomorphism,   sum_i s_i P_i = sum_i s1_i P_i + phi(sum_i s2_i P_i):  TWO MSMs with 128-bit scalars over
// the SAME points -- 7 windows of 19 / 18 bits instead of 13 of 20 / 19, so 7 window tables per basis instead of 13 -- and ONE
// application of phi, to the finished second sum, on the host (one Fq product).  No per-addition cost.
//
// lambda = z^2 - 1 for the BLS parameter z = -0xd201000000010000 (lambda^2 + lambda + 1 = r); beta is the cube root of unity in Fq
// with (beta x, y) = lambda (x, y) on the generator (checked numerically: tests/test_host_field.py, and by the byte parity of every
// MSM over an endomorphism SRS, tests/test_gpu_endo.py).  The split is a Barrett division by lambda: q ~ ((s >> 127) mu) >> 129 with
// mu = floor(2^256 / lambda) undershoots floor(s / lambda) by at most 2 (observed: 1), fixed by two conditional subtractions.
#pragma once
#include "field.hpp"
#include "g1.hpp"

namespace sonic {

#define ENDO_LAMBDA {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u}
#define ENDO_MU {0xf6cfee30u, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u}
#define ENDO_BETA_MONT {0x8671f071u, 0xcd03c9e4u, 0x1fcda5d2u, 0x5dab2246u, 0xd3851b95u, 0x587042afu, 0x01bacb9eu, 0x8eb60ebeu, 0x83d050d2u, 0x03f97d6eu, 0x54638741u, 0x18f02065u}
constexpr int ENDO_BITS = 130;          // what the windows of an endomorphism plan cover: 128 bits + room for the signed recoding's carry

// s (standard form, < r) -> s1 = s mod lambda, s2 = s div lambda, both as Fr-sized standard-form integers (upper four limbs zero)
HD void endo_split(const Fr& s, Fr& s1, Fr& s2) {
  constexpr uint32_t lam[4] = ENDO_LAMBDA;
  constexpr uint32_t mu[5] = ENDO_MU;
  // t = s >> 127 (128 bits)
  uint32_t t[4];
#pragma unroll
  for (int i = 0; i < 4; i++) t[i] = (s.l[3 + i] >> 31) | (s.l[4 + i] << 1);
  // p = t * mu (9 limbs); q = p >> 129
  uint32_t p[9];
#pragma unroll
  for (int i = 0; i < 9; i++) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      c += (uint64_t)t[i] * mu[j] + p[i + j];
      p[i + j] = (uint32_t)c;
      c >>= 32;
    }
    p[i + 5] = (uint32_t)c;
  }
  uint32_t q[4];
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = (p[4 + i] >> 1) | (p[5 + i] << 31);
  // rem = s - q * lambda: the true remainder is < 3 lambda < 2^130, so five limbs of the difference are enough
  uint32_t ql[5];
#pragma unroll
  for (int i = 0; i < 5; i++) ql[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i + j < 5) {
        c += (uint64_t)q[i] * lam[j] + ql[i + j];
        ql[i + j] = (uint32_t)c;
        c >>= 32;
      }
    }
    if (i + 4 < 5) ql[i + 4] = (uint32_t)c;
  }
  uint32_t rem[5];
  {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint64_t d = (uint64_t)s.l[i] - ql[i] - br; rem[i] = (uint32_t)d; br = (d >> 32) & 1; }
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    // rem >= lambda ?  (rem has 5 limbs, lambda 4)
    uint32_t d[5];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint64_t x = (uint64_t)rem[i] - (i < 4 ? lam[i] : 0u) - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
    if (!br) {
#pragma unroll
      for (int i = 0; i < 5; i++) rem[i] = d[i];
      uint64_t c = 1;
#pragma unroll
      for (int i = 0; i < 4; i++) { c += q[i]; q[i] = (uint32_t)c; c >>= 32; }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) { s1.l[i] = i < 4 ? rem[i] : 0u; s2.l[i] = i < 4 ? q[i] : 0u; }
}

// phi on an XYZZ point: x = X / ZZ, so (beta X, Y, ZZ, ZZZ)
HD G1XYZZ g1_endo(const G1XYZZ& p) {
  constexpr uint32_t b[12] = ENDO_BETA_MONT;
  Fq beta;
  for (int i = 0; i < 12; i++) beta.l[i] = b[i];
  G1XYZZ r = p;
  r.x = fp_mul(p.x, beta);
  return r;
}

}  // namespace sonic
