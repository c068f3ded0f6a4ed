// Lane-cooperative G1 arithmetic for the latency-bound ends of the bucket reduction (device only).
//
// A full XYZZ + XYZZ addition (add-2008-s) is 12 products + 2 squarings on ONE lane: ~9600 dependent instructions, ~19 us on a wave
// that has its SIMD to itself.  Where the reduction tree has fewer additions left than the chip has lanes, that chain is all that
// counts.  Here FOUR neighbouring lanes (a quad: lanes 4q .. 4q+3, the granule of DPP quad_perm) share one point addition: lane r
// holds coordinate r of every point (0: X, 1: Y, 2: ZZ, 3: ZZZ), the 14 products run as 5 rounds of one product per lane, and the
// operands move between the lanes with quad_perm DPP moves (12 per Fq value, full rate, no LDS):
//
//   round 1   U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1          (each lane: its P1 coordinate x the swapped P2 one)
//   round 2   PP = P^2    | RR = R^2     | ZZ1 ZZ2     | ZZZ1 ZZZ2             P = U2 - U1, R = S2 - S1
//   round 3   PPP = P PP  | Q = U1 PP    | ZZ3 = . PP  | (Q again)
//   round 4   --          | T = S1 PPP   | --          | ZZZ3 = . PPP          X3 = RR - PPP - 2Q on every lane
//   round 5   --          | R (Q - X3)   | --          | --                    Y3 = that - T
//
// 5 products deep instead of 14 (~2.6x shorter), 20 lane-products instead of 14 of issue: a trade that pays exactly where lanes idle.
// The doubling (dbl-2008-s-1) takes 4 rounds instead of 9 products.  Exceptional operands (infinity, equal x) are resolved with
// quad-uniform flags at the end; nothing branches per lane.
#pragma once
#include "g1.hpp"

namespace sonic {

#if defined(__HIPCC__)

// quad_perm control words: lane i of every quad reads lane sel[i]
constexpr int QP_SWAP2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);     // [2,3,0,1]
constexpr int QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA;          // broadcast of lane 0 / 1 / 2

template <int CTRL>
__device__ __forceinline__ Fq fq_quad(const Fq& a) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xf, 0xf, false);
  return r;
}
template <int CTRL>
__device__ __forceinline__ uint32_t u32_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ Fq fq_sel(bool c, const Fq& a, const Fq& b) {
  Fq r;
#pragma unroll
  for (int i = 0; i < 12; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// coordinate r (= lane & 3) of 2 P, from coordinate r of P
__device__ __forceinline__ Fq g1q_dbl(const Fq& a, const int r) {
  const uint32_t inf = u32_quad<QP_B2>(a.is_zero_strict() ? 1u : 0u) | u32_quad<QP_B1>(a.is_zero() ? 1u : 0u);   // ZZ = 0, or Y = 0 (order 2)
  const Fq t = fq_sel(r == 1, fp_dbl(a), a);                  // lane 1: U = 2Y
  const Fq m1 = fp_mul(t, t);                                 // XX | V = U^2 | - | -
  const Fq Vb = fq_quad<QP_B1>(m1);
  const Fq m2 = fp_mul(t, Vb);                                // S = X V | W = U V | ZZ3 = ZZ V | -
  const Fq Wb = fq_quad<QP_B1>(m2);
  const Fq Mb = fq_quad<QP_B0>(fp_add(fp_dbl(m1), m1));       // M = 3 XX
  const Fq m3 = fp_mul(fq_sel(r == 0, Mb, Wb), fq_sel(r == 0, Mb, a));     // M^2 | W Y | - | ZZZ3 = W ZZZ
  const Fq X3b = fq_quad<QP_B0>(fp_sub(m3, fp_dbl(m2)));      // X3 = M^2 - 2S
  const Fq Sb = fq_quad<QP_B0>(m2);
  const Fq m4 = fp_mul(Mb, fp_sub(Sb, X3b));                  // M (S - X3)
  Fq out = r == 0 ? X3b : r == 1 ? fp_sub(m4, m3) : r == 2 ? m2 : m3;
  if (inf) out = Fq::zero();
  return out;
}

// coordinate r of P1 + P2 (both XYZZ), from coordinate r of each
__device__ __forceinline__ Fq g1q_add(const Fq& a1, const Fq& a2, const int r) {
  const uint32_t inf1 = u32_quad<QP_B2>(a1.is_zero_strict() ? 1u : 0u), inf2 = u32_quad<QP_B2>(a2.is_zero_strict() ? 1u : 0u);
  const bool lo = r < 2;
  const Fq m1 = fp_mul(a1, fq_quad<QP_SWAP2>(a2));            // U1 | S1 | U2 | S2
  const Fq o = fq_quad<QP_SWAP2>(m1);                         // U2 | S2 | U1 | S1
  const Fq d = fp_sub(fq_sel(lo, o, m1), fq_sel(lo, m1, o));  // P | R | P | R
  const uint32_t dz = d.is_zero() ? 1u : 0u;
  const uint32_t pz = u32_quad<QP_B0>(dz), rz = u32_quad<QP_B1>(dz);
  const Fq m2 = fp_mul(fq_sel(lo, d, a1), fq_sel(lo, d, a2)); // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
  const Fq PPb = fq_quad<QP_B0>(m2);
  const Fq U1b = fq_quad<QP_B0>(m1);
  const Fq m3 = fp_mul(r == 0 ? d : r == 2 ? m2 : U1b, PPb);  // PPP | Q | ZZ3 | Q
  const Fq PPPb = fq_quad<QP_B0>(m3);
  const Fq Qb = fq_quad<QP_B1>(m3);
  const Fq X3 = fp_sub(fp_sub(fq_quad<QP_B1>(m2), PPPb), fp_dbl(Qb));
  const Fq m4 = fp_mul(fq_sel(r == 1, m1, m2), PPPb);         // - | T = S1 PPP | - | ZZZ3
  const Fq m5 = fp_mul(d, fp_sub(m3, X3));                    // - | R (Q - X3) | - | -
  Fq out = r == 0 ? X3 : r == 1 ? fp_sub(m5, m4) : r == 2 ? m3 : m4;
  // exceptional operands (flags are uniform over the quad): an operand at infinity returns the other one; equal x is a doubling
  // (equal y) or infinity (opposite y).  The doubling runs only if some quad of the wave needs it (wave-uniform branch).
  const bool same = !inf1 && !inf2 && pz;
  if (__any(same && rz)) {
    const Fq dbl = g1q_dbl(a1, r);
    if (same && rz) out = dbl;
  }
  if (same && !rz) out = Fq::zero();
  if (inf2) out = a1;
  if (inf1) out = a2;
  return out;
}

#endif  // device

}  // namespace sonic
