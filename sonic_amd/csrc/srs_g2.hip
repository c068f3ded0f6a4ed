// The G2 half of SRS.new (src/Sonic/SRS.hs:35-36,40-41):
//   basis 0: h^{x^e}       (hNegativeX[k] = e = -(k+1), hPositiveX[k] = e = k)
//   basis 1: h^{alpha x^e} (hNegativeAlphaX[k] = e = -(k+1), hPositiveAlphaX[k] = e = k)      e in [-d, d]
// The prover never reads these (the verifier reads hPositiveAlphaX[0], [1] and one h^{x^{-d+max}},
// CommitmentScheme.hs:58-68), so they are generated lazily, on the first sonic_srs_get_g2_points call: same
// scheme as the G1 side (byte table of the generator, 32 mixed additions per element, batched normalisation).
#include "internal.hpp"
#include "g2.hpp"

namespace sonic {

__device__ __forceinline__ G2Affine g2_generator() {
  constexpr uint32_t x0[12] = G2_GEN_X0_MONT, x1[12] = G2_GEN_X1_MONT, y0[12] = G2_GEN_Y0_MONT, y1[12] = G2_GEN_Y1_MONT;
  G2Affine g;
  for (int i = 0; i < 12; i++) { g.x.c0.l[i] = x0[i]; g.x.c1.l[i] = x1[i]; g.y.c0.l[i] = y0[i]; g.y.c1.l[i] = y1[i]; }
  return g;
}

// tab[w * 256 + j] = j * 2^(8w) * H (Jacobian); one thread per w
__global__ __launch_bounds__(64, 1) void k_g2_fb_table(G2Jac* __restrict__ tab) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 32) return;
  G2Affine h = g2_generator();
  G2Jac b; b.x = h.x; b.y = h.y; b.z = Fq2::one();
  for (int i = 0; i < 8 * w; i++) b = g2_dbl(b);
  const G2Affine base = g2_to_affine(b);
  G2Jac acc = G2Jac::inf();
  tab[w * 256] = acc;
  for (int j = 1; j < 256; j++) { acc = g2_add_mixed(acc, base); tab[w * 256 + j] = acc; }
}
__global__ __launch_bounds__(64, 1) void k_g2_to_affine(const G2Jac* __restrict__ in, G2Affine* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g2_to_affine(in[i]);
}

__device__ __forceinline__ G2Jac g2_fixed_base_mul(const G2Affine* __restrict__ tab, const Fr& k_mont) {
  Fr k = fp_from_mont(k_mont);
  G2Jac acc = G2Jac::inf();
#pragma unroll 1
  for (int w = 0; w < 32; w++) {
    uint32_t b = (k.l[w >> 2] >> (8 * (w & 3))) & 0xffu;
    if (b) acc = g2_add_mixed(acc, tab[w * 256 + b]);
  }
  return acc;
}

__global__ __launch_bounds__(256, 1) void k_g2_srs_points(const G2Affine* __restrict__ tab, long e_origin, long m, Fr x, Fr xinv, Fr alpha,
                                                          G2Jac* __restrict__ out0, G2Jac* __restrict__ out1) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  long e = i - e_origin;
  Fr p = e >= 0 ? fp_pow_u64(x, (uint64_t)e) : fp_pow_u64(xinv, (uint64_t)(-e));
  out0[i] = g2_fixed_base_mul(tab, p);
  out1[i] = g2_fixed_base_mul(tab, fp_mul(p, alpha));
}

// Montgomery's trick in Fq2 over chunks of 64 points
__global__ __launch_bounds__(64, 1) void k_g2_batch_affine(const G2Jac* __restrict__ in, G2Affine* __restrict__ out, Fq2* __restrict__ pref, long n) {
  constexpr int CH = 64;
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long lo = c * CH, hi = lo + CH;
  if (lo >= n) return;
  if (hi > n) hi = n;
  Fq2 acc = Fq2::one();
  for (long i = lo; i < hi; i++) {
    pref[i] = acc;
    G2Jac p = in[i];
    if (!p.is_inf()) acc = f2_mul(acc, p.z);
  }
  Fq2 inv = f2_inv(acc);
  for (long i = hi - 1; i >= lo; i--) {
    G2Jac p = in[i];
    if (p.is_inf()) { out[i] = G2Affine::inf(); continue; }
    Fq2 zi = f2_mul(inv, pref[i]);
    inv = f2_mul(inv, p.z);
    Fq2 zi2 = f2_sqr(zi);
    G2Affine a;
    a.x = f2_mul(p.x, zi2);
    a.y = f2_mul(p.y, f2_mul(zi2, zi));
    out[i] = a;
  }
}

__global__ void k_g2_setup_x(const Fr* in_std, Fr* out) {
  Fr x = fp_to_mont(in_std[0]), a = fp_to_mont(in_std[1]);
  out[0] = x; out[1] = fp_inv(x); out[2] = a;
}

__global__ __launch_bounds__(256) void k_g2_points_to_bytes(const G2Affine* __restrict__ in, uint8_t* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G2Affine p = in[i];
  uint32_t* w = reinterpret_cast<uint32_t*>(out + 192 * i);
  if (p.is_inf()) { for (int k = 0; k < 48; k++) w[k] = 0; return; }
  Fq a = fp_from_mont(p.x.c0), b = fp_from_mont(p.x.c1), c = fp_from_mont(p.y.c0), e = fp_from_mont(p.y.c1);
  for (int k = 0; k < 12; k++) { w[k] = a.l[k]; w[12 + k] = b.l[k]; w[24 + k] = c.l[k]; w[36 + k] = e.l[k]; }
}

// Caller-supplied G2 elements (an SRS file, sonic_srs_set_g2_points): canonical coordinates, on the twist
// y^2 = x^3 + 4(u + 1), and r P = O (E'(Fq2) has a large cofactor and the pairing is bilinear only on the order-r
// subgroup).  err bits as on the G1 side: 1 non-canonical, 2 off the curve, 4 outside the subgroup, 8 the point at infinity.
__global__ __launch_bounds__(64, 1) void k_g2_points_from_bytes(const uint8_t* __restrict__ in, G2Affine* __restrict__ out, long n, int* err) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(in + 192 * i);
  G2Affine p;
  for (int k = 0; k < 12; k++) { p.x.c0.l[k] = w[k]; p.x.c1.l[k] = w[12 + k]; p.y.c0.l[k] = w[24 + k]; p.y.c1.l[k] = w[36 + k]; }
  if (!fp_is_canonical(p.x.c0) || !fp_is_canonical(p.x.c1) || !fp_is_canonical(p.y.c0) || !fp_is_canonical(p.y.c1)) {
    atomicOr(err, 1); out[i] = G2Affine::inf(); return;
  }
  uint32_t nz = 0;
  for (int k = 0; k < 48; k++) nz |= w[k];
  // h^{x^e} and h^{alpha x^e} are never the identity (x, alpha != 0): a zero-filled section must not validate -- the verifier's
  // pairing with a G2 element at infinity is 1, and with all three at infinity it would accept every proof
  if (!nz) { atomicOr(err, 8); out[i] = G2Affine::inf(); return; }
  p.x.c0 = fp_to_mont(p.x.c0); p.x.c1 = fp_to_mont(p.x.c1); p.y.c0 = fp_to_mont(p.y.c0); p.y.c1 = fp_to_mont(p.y.c1);
  Fq2 b; b.c0 = fp_dbl(fp_dbl(Fq::one())); b.c1 = b.c0;
  if (!(f2_sqr(p.y) == f2_add(f2_mul(f2_sqr(p.x), p.x), b))) { atomicOr(err, 2); out[i] = G2Affine::inf(); return; }
  constexpr uint32_t rl[8] = FR_P;
  G2Jac acc; acc.x = p.x; acc.y = p.y; acc.z = Fq2::one();       // top bit (254) of r
#pragma unroll 1
  for (int bit = 253; bit >= 0; bit--) {
    acc = g2_dbl(acc);
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k == (bit >> 5)) word = rl[k];
    if ((word >> (bit & 31)) & 1u) acc = g2_add_mixed(acc, p);
  }
  if (!acc.is_inf()) { atomicOr(err, 4); out[i] = G2Affine::inf(); return; }
  out[i] = p;
}
void g2_points_from_bytes_enqueue(hipStream_t st, const uint8_t* d_in, G2Affine* out, long n, int* d_err) {
  if (n > 0) LAUNCH(k_g2_points_from_bytes, ceil_div(n, 64), 64, 0, st, d_in, out, n, d_err);
}

// fills h0 / h1 (2d+1 affine points each)
void srs_generate_g2(hipStream_t st, long d, const Fr& x_std, const Fr& alpha_std, G2Affine* h0, G2Affine* h1) {
  const long n = 2 * d + 1;
  DevBuf tabj(sizeof(G2Jac) * 8192), tab(sizeof(G2Affine) * 8192);
  LAUNCH(k_g2_fb_table, 1, 64, 0, st, tabj.as<G2Jac>());
  LAUNCH(k_g2_to_affine, ceil_div(8192, 64), 64, 0, st, (const G2Jac*)tabj.as<G2Jac>(), tab.as<G2Affine>(), 8192L);
  Fr h[2] = {x_std, alpha_std};
  DevBuf in(sizeof h), par(sizeof(Fr) * 3);
  HIP_OK(hipMemcpyAsync(in.p, h, sizeof h, hipMemcpyHostToDevice, st));
  LAUNCH(k_g2_setup_x, 1, 1, 0, st, (const Fr*)in.as<Fr>(), par.as<Fr>());
  Fr hp[3];
  HIP_OK(hipMemcpyAsync(hp, par.p, sizeof hp, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  const long SLAB = 1L << 19;
  const long cap = n < SLAB ? n : SLAB;
  DevBuf x0(sizeof(G2Jac) * cap), x1(sizeof(G2Jac) * cap), pref(sizeof(Fq2) * cap);
  for (long base = 0; base < n; base += SLAB) {
    long m = n - base < SLAB ? n - base : SLAB;
    LAUNCH(k_g2_srs_points, ceil_div(m, 256), 256, 0, st, (const G2Affine*)tab.as<G2Affine>(), d - base, m, hp[0], hp[1], hp[2],
           x0.as<G2Jac>(), x1.as<G2Jac>());
    LAUNCH(k_g2_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G2Jac*)x0.as<G2Jac>(), h0 + base, pref.as<Fq2>(), m);
    LAUNCH(k_g2_batch_affine, ceil_div(ceil_div(m, 64), 64), 64, 0, st, (const G2Jac*)x1.as<G2Jac>(), h1 + base, pref.as<Fq2>(), m);
  }
  HIP_OK(hipStreamSynchronize(st));
}

void g2_points_to_bytes_enqueue(hipStream_t st, const G2Affine* in, uint8_t* d_out, long n) {
  if (n > 0) LAUNCH(k_g2_points_to_bytes, ceil_div(n, 256), 256, 0, st, in, d_out, n);
}

}  // namespace sonic
