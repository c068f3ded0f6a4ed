// SRS.new on the GPU (src/Sonic/SRS.hs:27-43): the G1 vectors the prover reads,
//   basis 0: g^{x^e}, basis 1: g^{alpha x^e},  e in [-d, d]  (basis 1 has no e = 0 entry, SRS.hs:38)
// The reference computes every element as `mul gen (pow x i)` (a 255-bit double-and-add).  Here:
// a 32 x 256 table of byte multiples of the generator (j * 2^(8w) * G), one thread per exponent
// doing 2 x 32 table additions into XYZZ accumulators, then a batched (Montgomery-trick)
// normalisation to affine, 64 points per inversion.
#include "internal.hpp"
#include "endo.hpp"

namespace sonic {

__device__ __forceinline__ G1Affine g1_generator() {
  constexpr uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  G1Affine g;
  for (int i = 0; i < 12; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
  return g;
}

// tab[w * 256 + j] = j * 2^(8w) * G as XYZZ; one thread per w
__global__ __launch_bounds__(64) void k_fb_table(G1XYZZ* __restrict__ tab) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 32) return;
  G1XYZZ base = G1XYZZ::from_affine(g1_generator());
  for (int i = 0; i < 8 * w; i++) base = g1_dbl(base);
  G1XYZZ acc = G1XYZZ::inf();
  tab[w * 256] = acc;
  for (int j = 1; j < 256; j++) { acc = g1_add(acc, base); tab[w * 256 + j] = acc; }
}
__global__ __launch_bounds__(64) void k_xyzz_to_affine(const G1XYZZ* __restrict__ in, G1Affine* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g1_to_affine(in[i]);
}

__device__ __forceinline__ G1XYZZ fixed_base_mul(const G1Affine* __restrict__ tab, const Fr& k_mont) {
  Fr k = fp_from_mont(k_mont);
  G1XYZZ acc = G1XYZZ::inf();
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    uint32_t b = (k.l[w >> 2] >> (8 * (w & 3))) & 0xffu;
    if (b) acc = g1_add_mixed(acc, tab[w * 256 + b]);
  }
  return acc;
}

// slab slot i <-> exponent e = i - e_origin.  out0[i] = x^e * G, out1[i] = alpha x^e * G (XYZZ)
__global__ __launch_bounds__(256) void k_srs_points(const G1Affine* __restrict__ tab, long e_origin, long m, Fr x, Fr xinv, Fr alpha,
                                                    G1XYZZ* __restrict__ out0, G1XYZZ* __restrict__ out1) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  long e = i - e_origin;
  Fr p = e >= 0 ? fp_pow_u64(x, (uint64_t)e) : fp_pow_u64(xinv, (uint64_t)(-e));
  out0[i] = fixed_base_mul(tab, p);
  out1[i] = e == 0 ? G1XYZZ::inf() : fixed_base_mul(tab, fp_mul(p, alpha));
}

// Montgomery's trick over chunks of 64 points: one Fq inversion per chunk.
__global__ __launch_bounds__(64) void k_batch_affine(const G1XYZZ* __restrict__ in, PointArrayMut out, Fq* __restrict__ pref, long n) {
  constexpr int CH = 64;
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long lo = c * CH, hi = lo + CH;
  if (lo >= n) return;
  if (hi > n) hi = n;
  Fq acc = Fq::one();
  for (long i = lo; i < hi; i++) {
    pref[i] = acc;
    G1XYZZ p = in[i];
    if (!p.is_inf()) acc = fp_mul(acc, fp_mul(p.zz, p.zzz));
  }
  Fq inv = fp_inv(acc);
  for (long i = hi - 1; i >= lo; i--) {
    G1XYZZ p = in[i];
    if (p.is_inf()) { out[i] = G1Affine::inf(); continue; }
    Fq zi = fp_mul(inv, pref[i]);                 // 1 / (zz * zzz)
    inv = fp_mul(inv, fp_mul(p.zz, p.zzz));
    G1Affine a;
    a.x = fp_mul(p.x, fp_mul(zi, p.zzz));
    a.y = fp_mul(p.y, fp_mul(zi, p.zz));
    out[i] = a;
  }
}

// dst[i] = 2^c * src[i]
__global__ __launch_bounds__(256) void k_table_step(PointArray src, G1XYZZ* __restrict__ dst, long m, int c) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const G1Affine p = src[i];
  G1XYZZ acc = g1_dbl_affine(p);
  for (int k = 1; k < c; k++) acc = g1_dbl(acc);
  dst[i] = acc;
}

// ---- running sums of the alpha basis (round 5): ps[i] = A[0] + ... + A[i] in storage order (exponents -d .. i - d) ---------------
// commitPoly of a polynomial with a RUN of equal coefficients c over exponents [a, b] contributes c (A[a] + ... + A[b]) =
// c ps[b] - c ps[a - 1]: two terms instead of b - a + 1 (prove.hip, the unprepared S_j: rndCircuit's all-ones weight rows make 2n of
// the 3n + 1 coefficients of s(X, y_j) two values, test/Test/Reference.hs:141-155).  One more table of 2d + 1 affine points.
// Built slab by slab: serial sums inside 16-point chunks, a Hillis-Steele scan of the chunk totals, then every element takes the sum of
// the chunks (and slabs) before it.  The omitted g^alpha (infinity) adds nothing.
constexpr int PS_CHUNK = 16;
__global__ __launch_bounds__(64) void k_ps_chunk(PointArray src, G1XYZZ* __restrict__ x, G1XYZZ* __restrict__ tot, long m) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x, lo = c * PS_CHUNK;
  if (lo >= m) return;
  const long hi = lo + PS_CHUNK < m ? lo + PS_CHUNK : m;
  G1XYZZ acc = G1XYZZ::inf();
  for (long i = lo; i < hi; i++) { acc = g1_add_mixed(acc, src[i]); x[i] = acc; }
  tot[c] = acc;
}
__global__ __launch_bounds__(256) void k_ps_step(const G1XYZZ* __restrict__ in, G1XYZZ* __restrict__ out, long m, long off) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  out[i] = i >= off ? g1_add(in[i - off], in[i]) : in[i];
}
// x[i] += the inclusive scan of the chunk totals at the chunk before i's (+ the slabs before: carry)
__global__ __launch_bounds__(256) void k_ps_apply(G1XYZZ* __restrict__ x, const G1XYZZ* __restrict__ totscan, const G1XYZZ* __restrict__ carry, long m) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const long c = i / PS_CHUNK;
  G1XYZZ before = *carry;
  if (c > 0) before = g1_add(before, totscan[c - 1]);
  x[i] = g1_add(before, x[i]);
}

void srs_build_prefix(hipStream_t st, sonic_srs* s) {
  PointArrayMut ps = srs_prefix_mut(s);
  if (!ps.p) return;
  const long n = 2 * srs_d(s) + 1;
  const long SLAB = 1L << 20;
  const long cap = n < SLAB ? n : SLAB, ccap = ceil_div(cap, PS_CHUNK);
  DevBuf x(sizeof(G1XYZZ) * cap), pref(sizeof(Fq) * cap), t0(sizeof(G1XYZZ) * ccap), t1(sizeof(G1XYZZ) * ccap), carry(sizeof(G1XYZZ));
  HIP_OK(hipMemsetAsync(carry.p, 0, sizeof(G1XYZZ), st));            // the point at infinity
  const PointArray src = srs_basis(s, 1);
  for (long base = 0; base < n; base += SLAB) {
    const long m = n - base < SLAB ? n - base : SLAB, nc = ceil_div(m, PS_CHUNK);
    LAUNCH(k_ps_chunk, ceil_div(nc, 64), 64, 0, st, src + base, x.as<G1XYZZ>(), t0.as<G1XYZZ>(), m);
    G1XYZZ *a = t0.as<G1XYZZ>(), *b = t1.as<G1XYZZ>();
    for (long off = 1; off < nc; off <<= 1) {
      LAUNCH(k_ps_step, ceil_div(nc, 256), 256, 0, st, (const G1XYZZ*)a, b, nc, off);
      G1XYZZ* t = a; a = b; b = t;
    }
    LAUNCH(k_ps_apply, ceil_div(m, 256), 256, 0, st, x.as<G1XYZZ>(), (const G1XYZZ*)a, (const G1XYZZ*)carry.as<G1XYZZ>(), m);
    HIP_OK(hipMemcpyAsync(carry.p, x.as<G1XYZZ>() + (m - 1), sizeof(G1XYZZ), hipMemcpyDeviceToDevice, st));
    LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x.as<G1XYZZ>(), ps + base, pref.as<Fq>(), m);
  }
  HIP_OK(hipStreamSynchronize(st));
}

// ---- symmetric sums of the alpha basis (round 5): sym[e] = A[e] + A[-e] for e in [1, d], with window tables like a basis ------------
// s(u, Y) has the same coefficient -u^{i+n} at Y^i and at Y^{-i} (Constraints.hs:34-53 with X := u; poly.hip, k_s_of_u_diag), so the
// commitment C of hscProve (Signature.hs:51-52) is sum_i c_i (A[i] + A[-i]) + Q more terms: n terms over this table instead of 2n.
// Stored compactly (round 6): d + 1 points per window table, slot e for exponent e (slot 0 stays empty), the tables d + 1
// points apart -- the job over them shares a batched chain with the openings of the same polynomial and tells the kernels its own table
// stride (MsmJob::table_stride).  Up to round 5 the tables were laid out like a basis, 2d + 1 slots each, half of them never touched:
// 3.5 GB at d = 2^21 that were allocated, cleared and copied to every replica.
__global__ __launch_bounds__(256) void k_sym_sum(PointArray A, long d, long e0, long m, G1XYZZ* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const long e = e0 + i;
  out[i] = g1_add_mixed(G1XYZZ::from_affine(A[(size_t)(d + e)]), A[(size_t)(d - e)]);
}
void srs_build_sym(hipStream_t st, sonic_srs* s) {
  PointArrayMut sym = srs_sym_mut(s);
  if (!sym.p) return;
  const int W = srs_tab_W(s);
  const long d = srs_d(s), n = d + 1;                          // points per table: exponents 0 .. d
  for (int w = 0; w < W; w++) HIP_OK(hipMemsetAsync((sym + (long)((size_t)w * n)).p, 0, sym.stride, st));      // slot 0 of every table: infinity
  const long SLAB = 1L << 20;
  const long cap = d < SLAB ? d : SLAB;
  DevBuf x(sizeof(G1XYZZ) * cap), pref(sizeof(Fq) * cap);
  const PointArray A = srs_basis(s, 1);
  for (long base = 0; base < d; base += SLAB) {              // exponents e = 1 + base ...
    const long m = d - base < SLAB ? d - base : SLAB;
    LAUNCH(k_sym_sum, ceil_div(m, 256), 256, 0, st, A, d, 1 + base, m, x.as<G1XYZZ>());
    LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x.as<G1XYZZ>(), sym + (1 + base), pref.as<Fq>(), m);
  }
  for (int w = 1; w < W; w++)
    for (long base = 0; base < d; base += SLAB) {
      const long m = d - base < SLAB ? d - base : SLAB;
      LAUNCH(k_table_step, ceil_div(m, 256), 256, 0, st, (PointArray)(sym + (long)((size_t)(w - 1) * n + 1 + base)), x.as<G1XYZZ>(), m,
             msm_even_width(W, w - 1, srs_tab_endo(s) ? ENDO_BITS : 255));
      LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x.as<G1XYZZ>(), sym + (long)((size_t)w * n + 1 + base), pref.as<Fq>(), m);
    }
  HIP_OK(hipStreamSynchronize(st));
}

void srs_build_tables(hipStream_t st, sonic_srs* s) {
  srs_build_prefix(st, s);
  srs_build_sym(st, s);
  const int W = srs_tab_W(s), c = srs_tab_c(s);
  if (getenv("SONIC_DEBUG_TIMING")) fprintf(stderr, "[sonic] window tables: c=%d W=%d d=%ld\n", c, W, (long)srs_d(s));
  if (W <= 1) return;
  const long n = 2 * srs_d(s) + 1;
  const long SLAB = 1L << 20;
  const long cap = n < SLAB ? n : SLAB;
  DevBuf x(sizeof(G1XYZZ) * cap), pref(sizeof(Fq) * cap);
  for (int b = 0; b < 2; b++) {
    const PointArrayMut tab = srs_basis_mut(s, b);
    for (int w = 1; w < W; w++) {
      for (long base = 0; base < n; base += SLAB) {
        const long m = n - base < SLAB ? n - base : SLAB;
        LAUNCH(k_table_step, ceil_div(m, 256), 256, 0, st, (PointArray)(tab + (long)((size_t)(w - 1) * n + base)), x.as<G1XYZZ>(), m,
               msm_even_width(W, w - 1, srs_tab_endo(s) ? ENDO_BITS : 255));
        LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x.as<G1XYZZ>(), tab + (size_t)w * n + base, pref.as<Fq>(), m);
      }
    }
  }
  HIP_OK(hipStreamSynchronize(st));
}

__global__ void k_fr_setup_x(const Fr* in_std, Fr* out) {  // out = {x, x^-1, alpha} Montgomery
  Fr x = fp_to_mont(in_std[0]), a = fp_to_mont(in_std[1]);
  out[0] = x; out[1] = fp_inv(x); out[2] = a;
}

void srs_generate(hipStream_t st, sonic_srs* s, const Fr& x_std, const Fr& alpha_std) {
  const long d = srs_d(s), n = 2 * d + 1;
  DevBuf tabx(sizeof(G1XYZZ) * 8192), tab(sizeof(G1Affine) * 8192);
  LAUNCH(k_fb_table, 1, 64, 0, st, tabx.as<G1XYZZ>());
  LAUNCH(k_xyzz_to_affine, ceil_div(8192, 64), 64, 0, st, (const G1XYZZ*)tabx.as<G1XYZZ>(), tab.as<G1Affine>(), 8192L);
  Fr h[2] = {x_std, alpha_std};
  DevBuf in(sizeof h), par(sizeof(Fr) * 3);
  HIP_OK(hipMemcpyAsync(in.p, h, sizeof h, hipMemcpyHostToDevice, st));
  LAUNCH(k_fr_setup_x, 1, 1, 0, st, (const Fr*)in.as<Fr>(), par.as<Fr>());
  Fr hp[3];
  HIP_OK(hipMemcpyAsync(hp, par.p, sizeof hp, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  // process in slabs to bound the XYZZ scratch (2 x 192 B per slot)
  const long SLAB = 1L << 20;
  DevBuf x0(sizeof(G1XYZZ) * (n < SLAB ? n : SLAB)), x1(sizeof(G1XYZZ) * (n < SLAB ? n : SLAB)), pref(sizeof(Fq) * (n < SLAB ? n : SLAB));
  for (long base = 0; base < n; base += SLAB) {
    long m = n - base < SLAB ? n - base : SLAB;
    LAUNCH(k_srs_points, ceil_div(m, 256), 256, 0, st, (const G1Affine*)tab.as<G1Affine>(), d - base, m, hp[0], hp[1], hp[2],
           x0.as<G1XYZZ>(), x1.as<G1XYZZ>());
    LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x0.as<G1XYZZ>(), srs_basis_mut(s, 0) + base, pref.as<Fq>(), m);
    LAUNCH(k_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G1XYZZ*)x1.as<G1XYZZ>(), srs_basis_mut(s, 1) + base, pref.as<Fq>(), m);
  }
  HIP_OK(hipStreamSynchronize(st));
  srs_build_tables(st, s);
}

}  // namespace sonic
