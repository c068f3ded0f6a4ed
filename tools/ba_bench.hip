// Upper-bound experiment for BATCHED-AFFINE bucket accumulation on one MI355X: is "5 products + 1 squaring per
// addition behind a shared inversion" faster on this chip than the fused XYZZ mixed addition of k_bucket_accum (8M + 2S, 2608 MADs,
// measured 5.65e9 additions/s inside the kernel, 6.53e9/s from registers only)?
//
// One round of the pair tree, as a real implementation would run it: every thread owns B independent additions P1 + P2 of affine points.
//   pass 1  gather x1, x2 (the operands sit in the 1.26-GB window tables at random places, exactly like the entries of a bucket),
//           den = x2 - x1, running product, prefix stored to HBM (coalesced [i][thread])
//   inv     ONE inversion per thread and batch.  Stand-in: `inv_cost` dependent Montgomery products (a constant-time binary
//           GCD for 381 bits prices at ~46 products, Fermat at ~450) -- so what is printed is an UPPER bound on the speed
//   pass 2  operands gathered again (a thread cannot keep 50 x 192 B), prefix reloaded, inverse peeled off (2 products),
//           lambda, lambda^2, x3, y3 (2 products + 1 squaring), affine sum stored
// Outputs are garbage (the stand-in is not an inverse); the instruction mix, the register pressure and every byte of traffic are
// those of the real thing, minus the exceptional cases (equal x) and minus the compaction of the next round's operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sonic_amd/csrc tools/ba_bench.hip -o tools/ba_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "g1.hpp"
using namespace sonic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill(uint32_t* p, size_t words, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < words; i += stride) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    p[i] = (i % 12 == 11) ? (x & 0x0fffffffu) : x;                 // every 12-word element below 2^380 < q
  }
}
__global__ void k_idx(uint2* idx, size_t n, uint32_t T, int gather) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!gather) { idx[i] = make_uint2((uint32_t)((2 * i) % T), (uint32_t)((2 * i + 1) % T)); return; }   // later rounds: operands are neighbours in a compact array
  uint32_t a = (uint32_t)i * 0x9e3779b1u + 12345u, b = (uint32_t)i * 0x85ebca6bu + 54321u;
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; b ^= b >> 16; b *= 0x846ca68bu; b ^= b >> 13;
  idx[i] = make_uint2(a % T, b % T);
}

__global__ __launch_bounds__(256, 2) void k_ba_round(const G1Affine* __restrict__ pts, const uint2* __restrict__ idx, int B, Fq* __restrict__ prefix,
                                                     G1Affine* __restrict__ out, int inv_cost) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  Fq acc = Fq::one();
  {
    uint2 ij = idx[tid];
    Fq x1 = pts[ij.x].x, x2 = pts[ij.y].x;
    for (int i = 0; i < B; i++) {
      const int in = i + 1 < B ? i + 1 : i;
      const uint2 ijn = idx[(size_t)in * nth + tid];
      const Fq x1n = pts[ijn.x].x, x2n = pts[ijn.y].x;               // next pair in flight while this one is multiplied in
      prefix[(size_t)i * nth + tid] = acc;
      acc = fp_mul(acc, fp_sub(x2, x1));
      x1 = x1n; x2 = x2n;
    }
  }
  Fq inv = acc;
  for (int k = 0; k < inv_cost; k++) inv = fp_mul(inv, acc);          // stand-in for the one inversion of the batch
  {
    uint2 ij = idx[(size_t)(B - 1) * nth + tid];
    G1Affine p1 = pts[ij.x], p2 = pts[ij.y];
    Fq pre = prefix[(size_t)(B - 1) * nth + tid];
    for (int i = B - 1; i >= 0; i--) {
      const int in = i > 0 ? i - 1 : 0;
      const uint2 ijn = idx[(size_t)in * nth + tid];
      const G1Affine p1n = pts[ijn.x], p2n = pts[ijn.y];
      const Fq pren = prefix[(size_t)in * nth + tid];
      const Fq den = fp_sub(p2.x, p1.x);
      const Fq invi = fp_mul(inv, pre);
      inv = fp_mul(inv, den);
      const Fq lam = fp_mul(fp_sub(p2.y, p1.y), invi);
      G1Affine r;
      r.x = fp_sub(fp_sub(fp_sqr(lam), p1.x), p2.x);
      r.y = fp_sub(fp_mul(lam, fp_sub(p1.x, r.x)), p1.y);
      out[(size_t)i * nth + tid] = r;
      p1 = p1n; p2 = p2n; pre = pren;
    }
  }
}

// The fused XYZZ walk of k_bucket_accum on the same footing: every thread adds `B` table points, gathered at random places, into
// one accumulator, two-deep software pipeline as in the kernel.  `stride` = bytes between table points: 96 (packed, as the SRS tables
// are: half of the points straddle two 128-B lines) or 128 (one line per point, +33 % table memory).
__global__ __launch_bounds__(256, 2) void k_walk(const char* __restrict__ pts, const uint2* __restrict__ idx, int B, uint32_t stride, G1XYZZ* __restrict__ out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  auto pt = [&](uint32_t i) { return *reinterpret_cast<const G1Affine*>(pts + (size_t)i * stride); };
  uint32_t e_cur = idx[tid].x, e_nxt = idx[nth + tid].x;
  G1XYZZ acc = G1XYZZ::from_affine(pt(idx[tid].y));
  G1Affine p_cur = pt(e_cur);
  for (int i = 0; i < B; i++) {
    const uint32_t e_nn = idx[(size_t)(i + 2 < B ? i + 2 : B - 1) * nth + tid].x;
    const G1Affine p_nxt = pt(e_nxt);
    acc = g1_add_mixed_walk(acc, p_cur);
    p_cur = p_nxt; e_nxt = e_nn;
  }
  out[tid] = acc;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const uint32_t T = 13u << 20;                                       // 13 window tables of 2^20 points: 1.26 GB, the gather footprint of an N = 2^20 MSM
  const int threads = 256, blocks = pr.multiProcessorCount * 2;       // two workgroups per CU (what 2 waves per SIMD allow)
  const size_t nth = (size_t)blocks * threads;
  const int Bmax = 100;
  G1Affine* pts; CK(hipMalloc(&pts, (size_t)T * sizeof(G1Affine)));
  uint2* idx; CK(hipMalloc(&idx, nth * Bmax * sizeof(uint2)));
  Fq* prefix; CK(hipMalloc(&prefix, nth * Bmax * sizeof(Fq)));
  G1Affine* out; CK(hipMalloc(&out, nth * Bmax * sizeof(G1Affine)));
  hipLaunchKernelGGL(k_fill, 4096, 256, 0, 0, (uint32_t*)pts, (size_t)T * 24, 7u);
  CK(hipDeviceSynchronize());
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)k_ba_round));
  printf("device: %s CUs=%d; k_ba_round: %d VGPRs, %zu B scratch; %zu threads (2 workgroups of 256 per CU)\n", pr.name, pr.multiProcessorCount, fa.numRegs, (size_t)fa.localSizeBytes, nth);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int gather = 1; gather >= 0; gather--)
    for (int B : {12, 25, 50, 100})
      for (int inv_cost : {0, 46, 450}) {
        hipLaunchKernelGGL(k_idx, (unsigned)((nth * B + 255) / 256), 256, 0, 0, idx, nth * B, T, gather);
        hipLaunchKernelGGL(k_ba_round, blocks, threads, 0, 0, (const G1Affine*)pts, (const uint2*)idx, B, prefix, out, inv_cost);   // warm-up
        CK(hipDeviceSynchronize());
        const int reps = 3;
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_ba_round, blocks, threads, 0, 0, (const G1Affine*)pts, (const uint2*)idx, B, prefix, out, inv_cost);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double adds = (double)nth * B;
        // traffic by design per addition: pass 1 two 48-B x gathers + 8-B index + 48-B prefix store; pass 2 two 96-B point gathers + index + prefix + 96-B store
        const double bytes = adds * (2 * 48 + 8 + 48 + 2 * 96 + 8 + 48 + 96);
        printf("%s operands  B=%3d  inversion = %3d products: %8.3f ms  %.3e additions/s  (%.2f TB/s by design)\n", gather ? "gathered " : "sequential", B, inv_cost,
               ms, adds / (ms * 1e-3), bytes / (ms * 1e-3) / 1e12);
      }
  {
    // the same 13 x 2^20 table points at a 96-B and at a 128-B stride (the 128-B table reuses the first 3/4 of the points)
    const uint32_t T128 = (uint32_t)(((size_t)T * 96) / 128);
    for (uint32_t stride : {96u, 128u}) {
      const uint32_t Tn = stride == 96 ? T : T128;
      const int B = 26;
      hipLaunchKernelGGL(k_idx, (unsigned)((nth * B + 255) / 256), 256, 0, 0, idx, nth * B, Tn, 1);
      hipLaunchKernelGGL(k_walk, blocks, threads, 0, 0, (const char*)pts, (const uint2*)idx, B, stride, (G1XYZZ*)out);
      CK(hipDeviceSynchronize());
      const int reps = 5;
      hipEventRecord(e0, 0);
      for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_walk, blocks, threads, 0, 0, (const char*)pts, (const uint2*)idx, B, stride, (G1XYZZ*)out);
      hipEventRecord(e1, 0);
      CK(hipEventSynchronize(e1));
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("fused XYZZ walk, %u-B point stride, %d gathered points per thread: %8.3f ms  %.3e additions/s\n", stride, B, ms, (double)nth * B / (ms * 1e-3));
    }
  }
  printf("reference: fused XYZZ mixed addition 6.53e9 additions/s from registers only, 5.65e9/s inside k_bucket_accum (tools/microbench, bench.py)\n");
  return 0;
}
