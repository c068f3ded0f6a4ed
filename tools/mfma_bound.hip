// Upper bound for an MFMA-assisted Montgomery product on gfx950: does moving the constant-operand half of the
// product to the matrix pipe get past the 0.60 ceiling of the all-VALU instruction mix?
//
// Non-interleaved Montgomery product of 12-limb values:  T = a b  (24 limbs),  m = (T mod R) q' mod R,  t = (T + m q) / R.
// a b depends on two per-lane operands and stays on the VALU (144 v_mad_u64_u32 + their carry chains).  m and m q multiply a
// per-lane vector by a CONSTANT: over the 64 lanes of a wave that is a matrix product  Y = Toeplitz(c) X  with X the 48 (x 64 lanes)
// bytes of the per-lane operand -- genuine GEMM work for v_mfma_i32_32x32x32_i8:
//     m   = L(q') x   : 64 x 64 (k) x 64 lanes   ->  2 x 2 x 2 =  8 MFMAs   (output rows k < 48 padded to 64)
//     m q = L(q)  m   : 96 x 64 (k) x 64 lanes   ->  3 x 2 x 2 = 12 MFMAs
// What the VALU still has to do per product besides a b:
//   * operand layout: the B operand of a 32x32x32 tile wants lane l to hold bytes [16 (l / 32), +16) of column l % 32, the per-lane
//     vector lives whole in its own lane: one v_permlane32_swap per VGPR that has to cross the halves (12 in per GEMM)
//   * result layout: a lane gets 16 of the 32 rows of its column per tile, the other 16 sit in lane +-32: one swap per result VGPR
//     pair (32 for m, 48 for m q)
//   * i32 column sums (< 48 * 255^2 < 2^22) back to 32-bit limbs with carries: 4 columns per limb, ~6 instructions per limb
//     (72 for m, 144 for m q), + the signed-byte fix (i8 is signed: x ^ 0x80 per dword and a correction term)
// This file does NOT implement the exact arithmetic.  Like tools/ba_bench.hip it times a STAND-IN with the instruction mix of one
// product -- real v_mad_u64_u32 / carry chains for a b, real MFMAs on real register operands, real swaps, and the count of plain VALU
// instructions listed above, chained so that nothing is dead -- in a register-only loop, beside the all-VALU product it would replace
// (sonic_mont_mul_fq: 288 MADs + 362 others).  The stand-in leaves out what an exact version would add (the correction terms, the
// final conditional subtraction); its rate is therefore an upper bound.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "g1.hpp"
using namespace sonic;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// T = a b, 12 x 12 limbs -> 24: the VALU half (the compiler emits v_mad_u64_u32 + carry adds for this loop)
__device__ __forceinline__ void mul_12x12(const uint32_t* a, const uint32_t* b, uint32_t* t) {
#pragma unroll
  for (int i = 0; i < 24; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      c = (uint64_t)a[j] * b[i] + t[i + j] + c;
      t[i + j] = (uint32_t)c;
      c >>= 32;
    }
    t[i + 12] = (uint32_t)c;
  }
}

// `n` plain VALU instructions on x (a dependent add / xor / shift mix that cannot be folded)
template <int N>
__device__ __forceinline__ uint32_t valu_mix(uint32_t x, uint32_t y) {
#pragma unroll
  for (int i = 0; i < N; i += 3) { x += y; y ^= x >> 3; x = (x << 1) | (y & 1u); }
  return x ^ y;
}

// one product's worth of work in the MFMA formulation; `consts` = the Toeplitz tiles of q' and q as A operands (any bytes: a bound)
template <bool WITH_VALU_HALF, bool WITH_LAYOUT>
__global__ __launch_bounds__(256) void k_mfma_product(uint32_t* out, const v4i* consts, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[12], b[12];
  for (int i = 0; i < 12; i++) { a[i] = 0x9e3779b9u * (tid + i + 1); b[i] = 0x85ebca6bu * (tid + 7 * i + 3); }
  v4i A[4];
  for (int i = 0; i < 4; i++) A[i] = consts[(threadIdx.x & 63) * 4 + i];
  uint32_t sink = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t t[24];
    if (WITH_VALU_HALF) mul_12x12(a, b, t);
    else { for (int i = 0; i < 24; i++) t[i] = a[i % 12] ^ b[(i + 5) % 12]; }
    // ---- m = L(q') T_lo: B operands = the 12 low limbs (48 bytes) of every lane, 4 dwords per K-slab of 16 bytes
    v4i X[3];
    for (int s = 0; s < 3; s++) X[s] = (v4i){(int)(t[4 * s] ^ 0x80808080u), (int)(t[4 * s + 1] ^ 0x80808080u), (int)(t[4 * s + 2] ^ 0x80808080u), (int)(t[4 * s + 3] ^ 0x80808080u)};
    if (WITH_LAYOUT) {      // 12 dwords cross the lane halves
      for (int s = 0; s < 3; s++)
        for (int e = 0; e < 4; e++) {
          auto r = __builtin_amdgcn_permlane32_swap((unsigned)X[s][e], (unsigned)X[(s + 1) % 3][e], false, false);
          X[s][e] = (int)r[0];
        }
    }
    v16i D[4];
    for (int k = 0; k < 4; k++) D[k] = (v16i){0};
    // 2 (M tiles) x 2 (N tiles) x 2 (K steps) = 8 MFMAs
    for (int mt = 0; mt < 2; mt++)
      for (int nt = 0; nt < 2; nt++)
        for (int ks = 0; ks < 2; ks++)
          D[mt * 2 + nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(mt + ks) & 3], X[(nt + ks) % 3], D[mt * 2 + nt], 0, 0, 0);
    // result rows across the halves (32 swaps) and the column sums back into 12 limbs with carries (72 instructions)
    uint32_t m[12];
    if (WITH_LAYOUT) {
      for (int k = 0; k < 4; k++)
        for (int e = 0; e < 16; e += 2) {
          auto r = __builtin_amdgcn_permlane32_swap((unsigned)D[k][e], (unsigned)D[k][e + 1], false, false);
          D[k][e] = (int)r[0]; D[k][e + 1] = (int)r[1];
        }
    }
    for (int j = 0; j < 12; j++) m[j] = valu_mix<WITH_LAYOUT ? 6 : 1>((uint32_t)D[j & 3][j], (uint32_t)D[(j + 1) & 3][(j + 4) & 15]);
    // ---- m q = L(q) m: 3 x 2 x 2 = 12 MFMAs, B operands from m (12 swaps in), 48 swaps out, 144 instructions of carries
    v4i Y[3];
    for (int s = 0; s < 3; s++) Y[s] = (v4i){(int)(m[4 * s] ^ 0x80808080u), (int)(m[4 * s + 1] ^ 0x80808080u), (int)(m[4 * s + 2] ^ 0x80808080u), (int)(m[4 * s + 3] ^ 0x80808080u)};
    if (WITH_LAYOUT) {
      for (int s = 0; s < 3; s++)
        for (int e = 0; e < 4; e++) {
          auto r = __builtin_amdgcn_permlane32_swap((unsigned)Y[s][e], (unsigned)Y[(s + 1) % 3][e], false, false);
          Y[s][e] = (int)r[0];
        }
    }
    v16i E[6];
    for (int k = 0; k < 6; k++) E[k] = (v16i){0};
    for (int mt = 0; mt < 3; mt++)
      for (int nt = 0; nt < 2; nt++)
        for (int ks = 0; ks < 2; ks++)
          E[mt * 2 + nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[(mt + ks + 1) & 3], Y[(nt + ks) % 3], E[mt * 2 + nt], 0, 0, 0);
    if (WITH_LAYOUT) {
      for (int k = 0; k < 6; k++)
        for (int e = 0; e < 16; e += 2) {
          auto r = __builtin_amdgcn_permlane32_swap((unsigned)E[k][e], (unsigned)E[k][e + 1], false, false);
          E[k][e] = (int)r[0]; E[k][e + 1] = (int)r[1];
        }
    }
    // t = (T + m q) / R: 24 limbs of carry propagation (6 instructions each), the upper 12 are the result
    for (int j = 0; j < 12; j++) {
      const uint32_t lo = valu_mix<WITH_LAYOUT ? 6 : 1>((uint32_t)E[j % 6][j], t[j]);
      const uint32_t hi = valu_mix<WITH_LAYOUT ? 6 : 1>((uint32_t)E[(j + 3) % 6][(j + 5) & 15], t[12 + j]);
      a[j] = hi + (lo >> 31);
    }
    sink ^= a[0];
  }
  out[tid] = sink ^ a[3] ^ b[5];
}

// the all-VALU product this would replace (generated assembly, field.hpp)
__global__ __launch_bounds__(256) void k_valu_product(Fq* out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fq a, b;
  for (int i = 0; i < 12; i++) { a.l[i] = 0x9e3779b9u * (tid + i + 1); b.l[i] = 0x85ebca6bu * (tid + 7 * i + 3); }
  a.l[11] &= 0x0fffffffu; b.l[11] &= 0x0fffffffu;
  for (int it = 0; it < iters; it++) a = fp_mul(a, b);
  out[tid] = a;
}

template <class F> float time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("# tools/mfma_bound: MFMA-assisted Montgomery product, register-only loops on %d CUs (stand-in instruction mix: an UPPER BOUND, see the header)\n", pr.multiProcessorCount);
  void* buf; CK(hipMalloc(&buf, (size_t)pr.multiProcessorCount * 8 * 256 * sizeof(Fq)));
  v4i* consts; CK(hipMalloc((void**)&consts, 64 * 4 * sizeof(v4i)));
  { unsigned char h[64 * 4 * 16]; for (size_t i = 0; i < sizeof h; i++) h[i] = (unsigned char)(i * 37 + 11); CK(hipMemcpy(consts, h, sizeof h, hipMemcpyHostToDevice)); }
  const int it = 512;
  for (int wps = 1; wps <= 2; wps++) {               // waves per SIMD: 4 SIMDs per CU, 256-thread workgroups
    const int blocks = pr.multiProcessorCount * wps, threads = 256;
    const double lanes = (double)blocks * threads;
    const float v = time_ms([&] { hipLaunchKernelGGL(k_valu_product, blocks, threads, 0, 0, (Fq*)buf, it); }, 3);
    const float full = time_ms([&] { hipLaunchKernelGGL((k_mfma_product<true, true>), blocks, threads, 0, 0, (uint32_t*)buf, (const v4i*)consts, it); }, 3);
    const float nolay = time_ms([&] { hipLaunchKernelGGL((k_mfma_product<true, false>), blocks, threads, 0, 0, (uint32_t*)buf, (const v4i*)consts, it); }, 3);
    const float nomad = time_ms([&] { hipLaunchKernelGGL((k_mfma_product<false, true>), blocks, threads, 0, 0, (uint32_t*)buf, (const v4i*)consts, it); }, 3);
    const float mfma_only = time_ms([&] { hipLaunchKernelGGL((k_mfma_product<false, false>), blocks, threads, 0, 0, (uint32_t*)buf, (const v4i*)consts, it); }, 3);
    printf("%d wave(s) per SIMD: all-VALU product (generated assembly) %.3e products/s\n", wps, lanes * it / (v * 1e-3));
    printf("    MFMA form: a b on the VALU + 20 MFMAs + layout swaps + carry propagation   %.3e products/s  (%.2fx the all-VALU product)\n", lanes * it / (full * 1e-3), v / full);
    printf("    ... without the layout / carry instructions (a b + 20 MFMAs only)            %.3e products/s  (%.2fx)\n", lanes * it / (nolay * 1e-3), v / nolay);
    printf("    ... without a b (20 MFMAs + layout + carries)                                %.3e products/s  (%.2fx)\n", lanes * it / (nomad * 1e-3), v / nomad);
    printf("    ... the 20 MFMAs alone                                                       %.3e products/s  (%.2fx)\n", lanes * it / (mfma_only * 1e-3), v / mfma_only);
  }
  return 0;
}
