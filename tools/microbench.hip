// Microbenchmarks that size the integer roofline of the MSM on one MI355X:
//   mad     v_mad_u64_u32 issue rate (independent chains, all CUs)
//   fqmul   Montgomery products / s (the library's fp_mul<Fq>)
//   madd    XYZZ mixed additions / s
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sonic_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "g1.hpp"
#include "g1_quad.hpp"
using namespace sonic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mad(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc[8];
  for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = (uint64_t)x * y + acc[k];
    x += 3;
  }
  uint64_t s = 0;
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul32(uint32_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t acc[8];
  for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
  uint32_t y = b + threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = acc[k] * y + a;   // v_mad_u32_u24? no: v_mul_lo_u32 + add
    y += 3;
  }
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(double* out, double a, double b, int iters) {
  double acc[8];
  for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = __builtin_fma(acc[k], a, b);
  }
  double s = 0;
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add32(uint32_t* out, uint32_t a, int iters) {
  uint32_t acc[8];
  for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = (acc[k] + a) ^ (acc[k] >> 3);
  }
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fqmul(Fq* out, int iters) {
  Fq a = Fq::one(), b = Fq::r2();
  a.l[0] += threadIdx.x; b.l[1] ^= blockIdx.x;
  for (int i = 0; i < iters; i++) { a = fp_mul(a, b); b = fp_mul(b, a); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fp_add(a, b);
}
__global__ __launch_bounds__(256) void k_fqsqr(Fq* out, int iters) {
  Fq a = Fq::one(), b = Fq::r2();
  a.l[0] += threadIdx.x; b.l[1] ^= blockIdx.x;
  for (int i = 0; i < iters; i++) { a = fp_sqr(a); b = fp_sqr(b); }
  out[blockIdx.x * blockDim.x + threadIdx.x] = fp_add(a, b);
}
__global__ __launch_bounds__(256) void k_madd(G1XYZZ* out, const G1Affine* pts, int iters) {
  G1XYZZ acc = G1XYZZ::inf();
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; i++) acc = g1_add_mixed(acc, pts[(t + i * 7919) & 4095]);
  out[t] = acc;
}
__global__ __launch_bounds__(256) void k_madd_walk(G1XYZZ* out, const G1Affine* pts, int iters) {
  G1XYZZ acc = G1XYZZ::inf();
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; i++) acc = g1_add_mixed_walk(acc, pts[(t + i * 7919) & 4095]);
  out[t] = acc;
}
__global__ void k_mkpts(G1Affine* pts) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  G1Affine g; uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  for (int i = 0; i < 12; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
  G1XYZZ a = g1_mul_small(G1XYZZ::from_affine(g), (uint32_t)t + 1);
  pts[t] = g1_to_affine(a);
}

// asm product / sum / difference vs the C++ loops on the device: the canonical representative of the asm result must equal
// the C++ result bit for bit (Fq results live in [0, 2q): field.hpp "lazy range"), also for operands taken from [q, 2q)
template <class F> __device__ bool raw_eq(const F& a, const F& b) { uint32_t t = 0; for (int i = 0; i < F::N; i++) t |= a.l[i] ^ b.l[i]; return t == 0; }
template <class F> __device__ bool below_2p(const F& a) {
  F m = F::modulus(), d;                       // a - p must be < p
  uint64_t br = 0;
  for (int i = 0; i < F::N; i++) { uint64_t x = (uint64_t)a.l[i] - m.l[i] - br; d.l[i] = (uint32_t)x; br = (x >> 32) & 1; }
  return br || fp_is_canonical(d);
}
template <class F> __device__ F plus_p(const F& a) {       // a + p as raw limbs: the other representative of a
  F m = F::modulus(), r;
  uint64_t c = 0;
  for (int i = 0; i < F::N; i++) { c += (uint64_t)a.l[i] + m.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
  return r;
}
template <class F>
__global__ void k_check_mul(const F* in, int n, int* bad) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  F a = in[t], b = in[(t * 7 + 3) % n];
  auto same = [&](const F& x, const F& want) { if (!below_2p(x) || !raw_eq(fp_canonical(x), want)) atomicAdd(bad, 1); };
  same(fp_mul(a, b), fp_mul_generic(a, b));
  same(fp_mul(a, a), fp_mul_generic(a, a));
  same(fp_sqr(a), fp_mul_generic(a, a));          // Fq: the 78-product squaring routine
  same(fp_sqr(b), fp_mul_generic(b, b));
  same(fp_add(a, b), fp_add_generic(a, b));
  same(fp_sub(a, b), fp_sub_generic(a, b));
  same(fp_sub(b, a), fp_sub_generic(b, a));
  same(fp_add(a, a), fp_add_generic(a, a));
  same(fp_sub(a, a), fp_sub_generic(a, a));
  same(fp_neg(a), fp_sub_generic(F::zero(), a));
  if (fp_sub(a, a).is_zero() != true || (fp_mul(a, b) == fp_mul_generic(a, b)) != true) atomicAdd(bad, 1);
  if constexpr (F::N == 12) {                  // operands from the upper half of the lazy range
    F a2 = plus_p(a), b2 = plus_p(b);
    same(fp_mul(a2, b2), fp_mul_generic(a, b));
    same(fp_mul(a2, b), fp_mul_generic(a, b));
    same(fp_sqr(a2), fp_mul_generic(a, a));
    same(fp_sqr(b2), fp_mul_generic(b, b));
    same(fp_add(a2, b2), fp_add_generic(a, b));
    same(fp_sub(a2, b2), fp_sub_generic(a, b));
    same(fp_sub(a, b2), fp_sub_generic(a, b));
    same(fp_sub(a2, b), fp_sub_generic(a, b));
    same(fp_neg(a2), fp_sub_generic(F::zero(), a));
    if (!(a2 == a) || (a2 != a) || !fp_sub(a2, a).is_zero() || !raw_eq(fp_from_mont(a2), fp_from_mont(a))) atomicAdd(bad, 1);
  }
}
template <class F> int check_mul(const char* name) {
  const int n = 1 << 16;
  F* h = (F*)malloc(sizeof(F) * n);
  F pm = F::modulus();
  uint64_t st = 0x9e3779b97f4a7c15ull;
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < F::N; k++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i].l[k] = (uint32_t)(st >> 11); }
    h[i].l[F::N - 1] &= (F::N == 12 ? 0x0fffffffu : 0x3fffffffu);     // < p
    if (i < 8) {                                  // 0, 1, 2, 3 and p-1, p-2, p-3, p-4
      for (int k = 0; k < F::N; k++) h[i].l[k] = (i & 1) ? pm.l[k] : 0;
      if (i & 1) { uint64_t br = (i >> 1) + 1; for (int k = 0; k < F::N && br; k++) { uint64_t d = (uint64_t)h[i].l[k] - br; h[i].l[k] = (uint32_t)d; br = (d >> 32) & 1; } }
      else h[i].l[0] = i >> 1;
    }
    if (i >= 8 && i < 16) for (int k = 0; k < F::N; k++) h[i].l[k] = (k == F::N - 1) ? (pm.l[k] - 1) : 0xffffffffu;
  }
  F* d; int* bad; int hb = 0;
  hipMalloc(&d, sizeof(F) * n); hipMalloc(&bad, 4); hipMemcpy(d, h, sizeof(F) * n, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(k_check_mul<F>, n / 256, 256, 0, 0, (const F*)d, n, bad);
  hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("asm fp_mul<%s> vs C++ loop on %d pairs: %s (%d mismatches)\n", name, 2 * n, hb ? "FAIL" : "ok", hb);
  free(h); hipFree(d); hipFree(bad);
  return hb;
}

__global__ __launch_bounds__(256) void k_check_dbl(const G1Affine* pts, int n, int* bad, int c) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  G1Affine p = pts[t];
  G1XYZZ a = g1_dbl_affine(p);
  for (int k = 1; k < c; k++) a = g1_dbl(a);
  G1XYZZ b = g1_mul_small(G1XYZZ::from_affine(p), 1u << 11);
  G1Affine x = g1_to_affine(a), y = g1_to_affine(b);
  if (x.x != y.x || x.y != y.y) atomicAdd(bad, 1);
  // mixed add of equal points must take the doubling path
  G1XYZZ c2 = g1_add_mixed(G1XYZZ::from_affine(p), p);
  G1XYZZ d2 = g1_dbl(G1XYZZ::from_affine(p));
  G1Affine u = g1_to_affine(c2), v = g1_to_affine(d2);
  if (u.x != v.x || u.y != v.y) atomicAdd(bad + 1, 1);
  // the fused bucket-walk addition against the general one: general position (a few steps), and every exceptional position
  auto same = [&](const G1XYZZ& a, const G1XYZZ& b) {
    G1Affine s1 = g1_to_affine(a), s2 = g1_to_affine(b);
    if (a.is_inf() != b.is_inf() || s1.x != s2.x || s1.y != s2.y) atomicAdd(bad + 2, 1);
  };
  G1XYZZ w1 = a, w2 = a;
  for (int k = 0; k < 6; k++) {
    G1Affine q = pts[(t * 5 + 7 * k + 1) % n];
    if (k & 1) q = g1_neg(q);
    w1 = g1_add_mixed_walk(w1, q); w2 = g1_add_mixed(w2, q);
  }
  same(w1, w2);
  same(g1_add_mixed_walk(G1XYZZ::inf(), p), G1XYZZ::from_affine(p));            // accumulator at infinity
  same(g1_add_mixed_walk(a, G1Affine::inf()), a);                               // point at infinity
  same(g1_add_mixed_walk(G1XYZZ::from_affine(p), p), d2);                       // P + P
  same(g1_add_mixed_walk(d2, g1_to_affine(d2)), g1_dbl(d2));                    // P + P with a non-trivial ZZ
  same(g1_add_mixed_walk(G1XYZZ::from_affine(p), g1_neg(p)), G1XYZZ::inf());    // P + (-P)
  same(g1_add_mixed_walk(d2, g1_neg(g1_to_affine(d2))), G1XYZZ::inf());
  // half of the wave exceptional, half not (EXEC masking inside the statement)
  G1Affine qm = (t & 1) ? p : pts[(t + 3) % n];
  same(g1_add_mixed_walk(G1XYZZ::from_affine(p), qm), g1_add_mixed(G1XYZZ::from_affine(p), qm));
  // affine + affine (second entry of a walk): general position, P + P, P + (-P), infinity operands, "first only" lanes, mixed waves
  for (int k = 0; k < 4; k++) {
    G1Affine q = pts[(t * 3 + 11 * k + 2) % n];
    if (k & 1) q = g1_neg(q);
    same(g1_add_affine_walk(p, q, false), g1_add_mixed(G1XYZZ::from_affine(p), q));
  }
  same(g1_add_affine_walk(p, p, false), d2);
  same(g1_add_affine_walk(p, g1_neg(p), false), G1XYZZ::inf());
  same(g1_add_affine_walk(G1Affine::inf(), p, false), G1XYZZ::from_affine(p));
  same(g1_add_affine_walk(p, G1Affine::inf(), false), G1XYZZ::from_affine(p));
  same(g1_add_affine_walk(p, qm, true), G1XYZZ::from_affine(p));
  same(g1_add_affine_walk(p, qm, (t & 2) != 0), (t & 2) ? G1XYZZ::from_affine(p) : g1_add_mixed(G1XYZZ::from_affine(p), qm));
}

// the lane-cooperative addition / doubling (g1_quad.hpp) against the general ones: every quad takes its own case, so a wave mixes
// general position, P + P (doubling), P + (-P), either operand at infinity, both at infinity, and non-trivial ZZ on both sides
__global__ __launch_bounds__(256) void k_check_quad(const G1Affine* pts, int n, int* bad) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, q = t >> 2, r = t & 3;
  const G1Affine p = pts[q % n], o = pts[(q * 7 + 3) % n];
  G1XYZZ A = g1_dbl(g1_add_mixed(G1XYZZ::from_affine(p), o));         // some point with ZZ != 1
  G1XYZZ B = g1_add_mixed(g1_dbl_affine(o), p);
  switch (q % 8) {
    case 0: break;
    case 1: B = A; break;                                             // P + P
    case 2: B = g1_neg(A); break;                                     // P + (-P)
    case 3: A = G1XYZZ::inf(); break;
    case 4: B = G1XYZZ::inf(); break;
    case 5: A = G1XYZZ::inf(); B = G1XYZZ::inf(); break;
    case 6: B = g1_dbl(g1_dbl(A)); B = g1_add(B, g1_neg(g1_add(g1_dbl(A), A))); break;    // 4A - 3A: the SAME point as A in other coordinates
    default: A = G1XYZZ::from_affine(p); B = G1XYZZ::from_affine(o); break;             // ZZ = 1 on both sides
  }
  const Fq* ca = reinterpret_cast<const Fq*>(&A);
  const Fq* cb = reinterpret_cast<const Fq*>(&B);
  const Fq got = g1q_add(ca[r], cb[r], r);
  const G1XYZZ want = g1_add(A, B);
  // compare as group elements: gather the quad's four coordinates (every lane rebuilds the point)
  G1XYZZ G;
  Fq* cg = reinterpret_cast<Fq*>(&G);
  cg[0] = fq_quad<QP_B0>(got); cg[1] = fq_quad<QP_B1>(got); cg[2] = fq_quad<QP_B2>(got); cg[3] = fq_quad<0xFF>(got);
  const G1Affine x = g1_to_affine(G), y = g1_to_affine(want);
  if (G.is_inf() != want.is_inf() || x.x != y.x || x.y != y.y) atomicAdd(bad, 1);
  const Fq gd = g1q_dbl(ca[r], r);
  G1XYZZ D;
  Fq* cd = reinterpret_cast<Fq*>(&D);
  cd[0] = fq_quad<QP_B0>(gd); cd[1] = fq_quad<QP_B1>(gd); cd[2] = fq_quad<QP_B2>(gd); cd[3] = fq_quad<0xFF>(gd);
  const G1XYZZ wd = g1_dbl(A);
  const G1Affine u = g1_to_affine(D), v = g1_to_affine(wd);
  if (D.is_inf() != wd.is_inf() || u.x != v.x || u.y != v.y) atomicAdd(bad + 1, 1);
}
// dependent chains of additions: one lane per chain (general addition) and one quad per chain -- the latency the reduction tree pays
__global__ __launch_bounds__(256) void k_chain_full(G1XYZZ* out, const G1Affine* pts, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  G1XYZZ acc = g1_dbl_affine(pts[t & 4095]);
  const G1XYZZ inc = g1_dbl_affine(pts[(t * 3 + 1) & 4095]);
  for (int i = 0; i < iters; i++) acc = g1_add(acc, inc);
  out[t] = acc;
}
__global__ __launch_bounds__(256) void k_chain_quad(G1XYZZ* out, const G1Affine* pts, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, q = t >> 2, r = t & 3;
  const G1XYZZ a0 = g1_dbl_affine(pts[q & 4095]), i0 = g1_dbl_affine(pts[(q * 3 + 1) & 4095]);
  Fq acc = reinterpret_cast<const Fq*>(&a0)[r];
  const Fq inc = reinterpret_cast<const Fq*>(&i0)[r];
  for (int i = 0; i < iters; i++) acc = g1q_add(acc, inc, r);
  reinterpret_cast<Fq*>(&out[q])[r] = acc;
}

// NTT butterflies from registers only (no LDS, no memory): what the stages of ntt.hip could reach if nothing but issue counted
__global__ __launch_bounds__(256) void k_butterfly(Fr* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fr a, b, w;
  for (int i = 0; i < 8; i++) { a.l[i] = 0x9e3779b9u * (t + i + 1); b.l[i] = 0x85ebca6bu * (t + 3 * i + 7); w.l[i] = 0xc2b2ae35u * (t + 5 * i + 11); }
  a.l[7] &= 0x3fffffffu; b.l[7] &= 0x3fffffffu; w.l[7] &= 0x3fffffffu;
  for (int i = 0; i < iters; i++) { const Fr s = fp_add(a, b); b = fp_mul(fp_sub(a, b), w); a = s; }
  out[t] = fp_add(a, b);
}

template <class F> float time_ms(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv) {
  const bool check_only = argc > 1 && !strcmp(argv[1], "--check");      // self-checks only (tests/test_gpu_asm.py)
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  printf("device: %s CUs=%d clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  if (check_mul<Fq>("Fq") | check_mul<Fr>("Fr")) return 2;
  const int blocks = pr.multiProcessorCount * 8, threads = 256;
  void* buf; CK(hipMalloc(&buf, (size_t)blocks * threads * sizeof(G1XYZZ)));
  G1Affine* pts; CK(hipMalloc(&pts, 4096 * sizeof(G1Affine)));
  hipLaunchKernelGGL(k_mkpts, 16, 256, 0, 0, pts); CK(hipDeviceSynchronize());
  { int* bad; int hb[3] = {0, 0, 0}; hipMalloc(&bad, 12); hipMemset(bad, 0, 12);
    hipLaunchKernelGGL(k_check_dbl, 16, 256, 0, 0, (const G1Affine*)pts, 4096, bad, 11); hipMemcpy(hb, bad, 12, hipMemcpyDeviceToHost);
    printf("dbl_affine^11 vs mul_small(2^11): %d mismatches; add_mixed(P,P) vs dbl: %d mismatches; fused walk addition vs general (incl. exceptional lanes): %d mismatches\n", hb[0], hb[1], hb[2]);
    if (hb[0] | hb[1] | hb[2]) return 3; }
  { int* bad; int hb[2] = {0, 0}; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_check_quad, 64, 256, 0, 0, (const G1Affine*)pts, 4096, bad); hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("quad addition vs general (8 cases per wave, exceptional ones included): %d mismatches; quad doubling: %d mismatches\n", hb[0], hb[1]);
    if (hb[0] | hb[1]) return 4; }
  if (check_only) { printf("self-checks ok\n"); return 0; }
  const double lanes = (double)blocks * threads;
  { int it = 32768; float ms = time_ms([&] { hipLaunchKernelGGL(k_mad, blocks, threads, 0, 0, (uint64_t*)buf, 12345u, 67891u, it); }, 10);
    printf("v_mad_u64_u32: %.3f ms -> %.3e mad/s  (%.2f lane-ops/clk/CU at 2.4GHz)\n", ms, lanes * it * 8 / (ms * 1e-3), lanes * it * 8 / (ms * 1e-3) / 2.4e9 / pr.multiProcessorCount); }
  { int it = 32768; float ms = time_ms([&] { hipLaunchKernelGGL(k_mul32, blocks, threads, 0, 0, (uint32_t*)buf, 12345u, 67891u, it); }, 10);
    printf("mul_lo+add:    %.3f ms -> %.3e op/s  (%.2f lane-ops/clk/CU)\n", ms, lanes * it * 8 / (ms * 1e-3), lanes * it * 8 / (ms * 1e-3) / 2.4e9 / pr.multiProcessorCount); }
  { int it = 32768; float ms = time_ms([&] { hipLaunchKernelGGL(k_fma64, blocks, threads, 0, 0, (double*)buf, 1.0000001, 0.5, it); }, 10);
    printf("v_fma_f64:     %.3f ms -> %.3e fma/s (%.2f lane-ops/clk/CU)\n", ms, lanes * it * 8 / (ms * 1e-3), lanes * it * 8 / (ms * 1e-3) / 2.4e9 / pr.multiProcessorCount); }
  { int it = 32768; float ms = time_ms([&] { hipLaunchKernelGGL(k_add32, blocks, threads, 0, 0, (uint32_t*)buf, 12345u, it); }, 10);
    printf("add+xor+shift: %.3f ms -> %.3e triple/s (%.2f lane-triples/clk/CU)\n", ms, lanes * it * 8 / (ms * 1e-3), lanes * it * 8 / (ms * 1e-3) / 2.4e9 / pr.multiProcessorCount); }
  { int it = 256; float ms = time_ms([&] { hipLaunchKernelGGL(k_fqmul, blocks, threads, 0, 0, (Fq*)buf, it); }, 3);
    printf("fq_mul:        %.3f ms -> %.3e mul/s\n", ms, lanes * it * 2 / (ms * 1e-3)); }
  { int it = 256; float ms = time_ms([&] { hipLaunchKernelGGL(k_fqsqr, blocks, threads, 0, 0, (Fq*)buf, it); }, 3);
    printf("fq_sqr:        %.3f ms -> %.3e sqr/s\n", ms, lanes * it * 2 / (ms * 1e-3)); }
  { int it = 64; float ms = time_ms([&] { hipLaunchKernelGGL(k_madd, blocks, threads, 0, 0, (G1XYZZ*)buf, (const G1Affine*)pts, it); }, 3);
    printf("g1_add_mixed:  %.3f ms -> %.3e add/s\n", ms, lanes * it / (ms * 1e-3)); }
  { int it = 64; float ms = time_ms([&] { hipLaunchKernelGGL(k_madd_walk, blocks, threads, 0, 0, (G1XYZZ*)buf, (const G1Affine*)pts, it); }, 3);
    printf("g1_add_mixed_walk (fused asm): %.3f ms -> %.3e add/s\n", ms, lanes * it / (ms * 1e-3)); }
  // the same register-only walk addition with exactly ONE and exactly TWO waves resident per SIMD (one / two 256-lane workgroups per CU): what a
  // second independent instruction stream per SIMD is worth -- the most that interleaving two buckets inside one thread could give at one wave
  // per SIMD (round 6; tools/accum_isa.py has the issue-bound of the instruction mix)
  for (int wps = 1; wps <= 2; wps++) {
    int it = 256; const int bl = pr.multiProcessorCount * wps, th = 256;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_madd_walk, bl, th, 0, 0, (G1XYZZ*)buf, (const G1Affine*)pts, it); }, 3);
    printf("g1_add_mixed_walk from registers, %d wave(s) per SIMD: %.3e add/s (%.2f us per wave-addition on a SIMD)\n", wps, (double)bl * th * it / (ms * 1e-3),
           1e3 * ms / it / wps);
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    int it = 2048; const int bl = pr.multiProcessorCount * wps, th = 256;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_butterfly, bl, th, 0, 0, (Fr*)buf, it); }, 3);
    printf("Fr butterfly (add, sub, product) from registers, %d wave(s) per SIMD: %.3e butterflies/s (a stage of 2^20 in %.1f us)\n", wps, (double)bl * th * it / (ms * 1e-3), 1048576.0 / ((double)bl * th * it / (ms * 1e-3)) * 1e6);
  }
  // latency of a dependent addition: ONE wave per SIMD (the late levels of the bucket-reduction tree), whole addition per lane vs per quad
  { int it = 32; const int bl = pr.multiProcessorCount, th = 256;
    float m1 = time_ms([&] { hipLaunchKernelGGL(k_chain_full, bl, th, 0, 0, (G1XYZZ*)buf, (const G1Affine*)pts, it); }, 3);
    float m2 = time_ms([&] { hipLaunchKernelGGL(k_chain_quad, bl, th, 0, 0, (G1XYZZ*)buf, (const G1Affine*)pts, it); }, 3);
    printf("dependent XYZZ addition, one wave per SIMD: %.2f us per addition on a lane, %.2f us on a quad (%.2fx)\n", 1e3 * m1 / it, 1e3 * m2 / it, m1 / m2); }
  return 0;
}
