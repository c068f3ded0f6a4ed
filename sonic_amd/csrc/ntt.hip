// NTT over Fr (radix 2 butterflies, the wide stages two per memory pass) for the Laurent-polynomial product t(X,y) = r(X,1) * (r(X,y) + s(X,y))
// (src/Sonic/Constraints.hs:61, where poly-0.4.0.0 runs a sparse convolution whose coefficients
// are themselves polynomials; evaluating Y := y first is a ring homomorphism, so the univariate
// product is result-exact).  omega_n = 7^((r-1)/n); Fr has 2-adicity 32.
//
// Forward = decimation in frequency (natural in, bit-reversed out); inverse = decimation in time
// (bit-reversed in, natural out, scaled by 1/n): the product needs no permutation pass.
// Stages whose butterfly span fits a 2048-element tile (64 KB of the CU's 160 KB LDS) run fused in
// one kernel out of LDS; wider stages stream through HBM one stage per launch.
#include <algorithm>
#include "internal.hpp"

namespace sonic {

static constexpr int TILE_LOG = 11;

__device__ __forceinline__ Fr root_2_32(bool inverse) {
  constexpr uint32_t w[8] = FR_ROOT_2_32_MONT;
  constexpr uint32_t wi[8] = FR_ROOT_2_32_INV_MONT;
  Fr r;
  for (int i = 0; i < 8; i++) r.l[i] = inverse ? wi[i] : w[i];
  return r;
}

// tw[k] = w^k, k < half, w the primitive 2^log2n-th root (or its inverse)
__global__ __launch_bounds__(256) void k_ntt_twiddles(Fr* __restrict__ tw, long half, int log2n, int inverse) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long k0 = t * 16;
  if (k0 >= half) return;
  Fr w = root_2_32(inverse);
  for (int i = log2n; i < FR_TWO_ADICITY; i++) w = fp_sqr(w);
  Fr p = fp_pow_u64(w, (uint64_t)k0);
  for (int j = 0; j < 16 && k0 + j < half; j++) { tw[k0 + j] = p; p = fp_mul(p, w); }
}

// one butterfly per thread, stage s of a DIF (forward) or DIT (inverse) pass through HBM
// The streaming kernels below run grid-stride over a capped grid (WIDE_GRID workgroups).  Launched beside a bucket accumulation,
// which holds every wave slot with long-lived waves, a kernel gets a slot only when an accumulation workgroup retires: with one
// short workgroup per 256 elements a 2^23-point stage needed 8192 such grants and took 36 ms instead of 0.15 ms; a few hundred
// long-lived workgroups need a few hundred.
static constexpr int WIDE_GRID = 512;
static inline int wide_grid(long items) { long g = (items + 255) / 256; return (int)(g < WIDE_GRID ? g : WIDE_GRID); }

__global__ __launch_bounds__(256) void k_ntt_stage(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s, int tw_shift, int inverse) {
  const long halfn = 1L << (log2n - 1);
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < halfn; t += (long)gridDim.x * blockDim.x) {
  long half = 1L << (log2n - 1 - s);
  long j = t & (half - 1);
  long i0 = ((t >> (log2n - 1 - s)) << (log2n - s)) + j;
  long i1 = i0 + half;
  Fr w = tw[(j << s) << tw_shift];
  Fr a = d[i0], b = d[i1];
  if (!inverse) { d[i0] = fp_add(a, b); d[i1] = fp_mul(fp_sub(a, b), w); }
  else { Fr bw = fp_mul(b, w); d[i0] = fp_add(a, bw); d[i1] = fp_sub(a, bw); }
  }
}

// Two consecutive wide stages in one pass through HBM (radix 4): a thread owns x[i0 + k q], k = 0..3, q = n >> (s + 2), and does
// stage s (span 2q) and stage s + 1 (span q) on them -- DIF order forward, the reverse (DIT) order inverse.  Halves the number of
// 64-B-per-element passes of the wide part (2^23 points: 12 -> 6 passes per transform).
__global__ __launch_bounds__(256) void k_ntt_stage2(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s, int tw_shift, int inverse) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (1L << (log2n - 2)); t += (long)gridDim.x * blockDim.x) {
  const int lq = log2n - 2 - s;                    // log2 q
  const long q = 1L << lq;
  const long j = t & (q - 1);
  const long i0 = ((t >> lq) << (lq + 2)) + j;
  Fr x0 = d[i0], x1 = d[i0 + q], x2 = d[i0 + 2 * q], x3 = d[i0 + 3 * q];
  const Fr wa = tw[(j << s) << tw_shift];          // stage s, position j
  const Fr wb = tw[((j + q) << s) << tw_shift];    // stage s, position j + q
  const Fr wc = tw[(j << (s + 1)) << tw_shift];    // stage s + 1, position j
  if (!inverse) {
    const Fr y0 = fp_add(x0, x2), y2 = fp_mul(fp_sub(x0, x2), wa);
    const Fr y1 = fp_add(x1, x3), y3 = fp_mul(fp_sub(x1, x3), wb);
    d[i0] = fp_add(y0, y1);
    d[i0 + q] = fp_mul(fp_sub(y0, y1), wc);
    d[i0 + 2 * q] = fp_add(y2, y3);
    d[i0 + 3 * q] = fp_mul(fp_sub(y2, y3), wc);
  } else {
    const Fr b1 = fp_mul(x1, wc), b3 = fp_mul(x3, wc);
    const Fr y0 = fp_add(x0, b1), y1 = fp_sub(x0, b1);
    const Fr y2 = fp_add(x2, b3), y3 = fp_sub(x2, b3);
    const Fr c2 = fp_mul(y2, wa), c3 = fp_mul(y3, wb);
    d[i0] = fp_add(y0, c2);
    d[i0 + 2 * q] = fp_sub(y0, c2);
    d[i0 + q] = fp_add(y1, c3);
    d[i0 + 3 * q] = fp_sub(y1, c3);
  }
  }
}

// all stages with span <= tile, fused in LDS.  tile_log = min(log2n, TILE_LOG).
__global__ __launch_bounds__(256) void k_ntt_local(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                                   int inverse, const Fr* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Fr* sh = reinterpret_cast<Fr*>(smem);
  const int tile = 1 << tile_log;
  const long ntiles = 1L << (log2n - tile_log);
  for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {       // grid-stride over the tiles (see WIDE_GRID)
  const long base = tl << tile_log;
  for (int i = threadIdx.x; i < tile; i += 256) sh[i] = d[base + i];
  __syncthreads();
  // forward: stages s = log2n - tile_log .. log2n - 1 (half = tile/2 .. 1)
  // inverse: the same stages in reverse order (half = 1 .. tile/2)
  for (int k = 0; k < tile_log; k++) {
    const int hl = inverse ? k : tile_log - 1 - k;       // log2(half)
    const int s = log2n - 1 - hl;
    for (int bt = threadIdx.x; bt < tile / 2; bt += 256) {
      int j = bt & ((1 << hl) - 1);
      int i0 = ((bt >> hl) << (hl + 1)) + j;
      int i1 = i0 + (1 << hl);
      Fr w = tw[((long)j << s) << tw_shift];
      Fr a = sh[i0], b = sh[i1];
      if (!inverse) { sh[i0] = fp_add(a, b); sh[i1] = fp_mul(fp_sub(a, b), w); }
      else { Fr bw = fp_mul(b, w); sh[i0] = fp_add(a, bw); sh[i1] = fp_sub(a, bw); }
    }
    __syncthreads();
  }
  if (scale) { Fr sc = *scale; for (int i = threadIdx.x; i < tile; i += 256) d[base + i] = fp_mul(sh[i], sc); }
  else { for (int i = threadIdx.x; i < tile; i += 256) d[base + i] = sh[i]; }
  __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_fr_scale(Fr* __restrict__ a, long n, const Fr* __restrict__ s) {
  const Fr sc = *s;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = fp_mul(a[i], sc);
}
void fr_scale_enqueue(hipStream_t st, Fr* a, long n, const Fr* s) { LAUNCH(k_fr_scale, wide_grid(n), 256, 0, st, a, n, s); }

__global__ void k_fr_inv_pow2(Fr* out) {   // out[k] = (2^k)^-1 in Montgomery form, k = 0..32
  int k = threadIdx.x;
  if (k > 32) return;
  Fr two = fp_dbl(Fr::one());
  out[k] = fp_inv(fp_pow_u64(two, (uint64_t)k));
}

void NttTables::ensure(hipStream_t st, int need) {
  if (need <= log2n) return;
  HIP_OK(hipStreamSynchronize(st));   // earlier launches may still read the old tables
  long half = 1L << (need - 1);
  fwd.alloc(sizeof(Fr) * half);
  inv.alloc(sizeof(Fr) * half);
  LAUNCH(k_ntt_twiddles, ceil_div(ceil_div(half, 16), 256), 256, 0, st, fwd.as<Fr>(), half, need, 0);
  LAUNCH(k_ntt_twiddles, ceil_div(ceil_div(half, 16), 256), 256, 0, st, inv.as<Fr>(), half, need, 1);
  if (!ninv.p) { ninv.alloc(sizeof(Fr) * 33); LAUNCH(k_fr_inv_pow2, 1, 64, 0, st, ninv.as<Fr>()); }
  log2n = need;
}

static void ntt_run(hipStream_t st, const NttTables& tw, Fr* d, int log2n, bool inverse) {
  if (log2n == 0) return;
  const int tile_log = log2n < TILE_LOG ? log2n : TILE_LOG;
  const int tw_shift = tw.log2n - log2n;
  const Fr* table = (inverse ? tw.inv : tw.fwd).as<Fr>();
  const long n = 1L << log2n;
  const int nglobal = log2n - tile_log;
  const size_t lds = sizeof(Fr) << tile_log;
  if (!inverse) {
    int s = 0;
    for (; s + 1 < nglobal; s += 2) LAUNCH(k_ntt_stage2, wide_grid(n / 4), 256, 0, st, d, table, log2n, s, tw_shift, 0);
    for (; s < nglobal; s++) LAUNCH(k_ntt_stage, wide_grid(n / 2), 256, 0, st, d, table, log2n, s, tw_shift, 0);
    LAUNCH(k_ntt_local, (int)std::min<long>(n >> tile_log, WIDE_GRID), 256, lds, st, d, table, log2n, tile_log, tw_shift, 0, (const Fr*)nullptr);
  } else {
    const Fr* ninv = tw.ninv.as<Fr>() + log2n;
    if (nglobal == 0) {
      LAUNCH(k_ntt_local, (int)std::min<long>(n >> tile_log, WIDE_GRID), 256, lds, st, d, table, log2n, tile_log, tw_shift, 1, ninv);
    } else {
      LAUNCH(k_ntt_local, (int)std::min<long>(n >> tile_log, WIDE_GRID), 256, lds, st, d, table, log2n, tile_log, tw_shift, 1, (const Fr*)nullptr);
      int s = nglobal - 1;
      if (nglobal & 1) { LAUNCH(k_ntt_stage, wide_grid(n / 2), 256, 0, st, d, table, log2n, s, tw_shift, 1); s--; }   // the odd one first: the forward pass did it last
      for (; s >= 1; s -= 2) LAUNCH(k_ntt_stage2, wide_grid(n / 4), 256, 0, st, d, table, log2n, s - 1, tw_shift, 1);
      fr_scale_enqueue(st, d, n, ninv);
    }
  }
}

void ntt_forward_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, false); }
void ntt_inverse_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, true); }

__global__ __launch_bounds__(256) void k_fr_pointwise_mul(Fr* __restrict__ a, const Fr* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = fp_mul(a[i], b[i]);
}
void fr_pointwise_mul_enqueue(hipStream_t st, Fr* a, const Fr* b, long n) { LAUNCH(k_fr_pointwise_mul, wide_grid(n), 256, 0, st, a, b, n); }

__global__ __launch_bounds__(256) void k_fr_bitrev(Fr* __restrict__ d, int log2n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1L << log2n)) return;
  long j = (long)(__brevll((unsigned long long)i) >> (64 - log2n));
  if (i < j) { Fr a = d[i], b = d[j]; d[i] = b; d[j] = a; }
}
void fr_bitrev_permute_enqueue(hipStream_t st, Fr* d, int log2n) {
  if (log2n > 0) LAUNCH(k_fr_bitrev, ceil_div(1L << log2n, 256), 256, 0, st, d, log2n);
}

}  // namespace sonic
