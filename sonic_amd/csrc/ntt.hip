// Radix-2 NTT over Fr for the Laurent-polynomial product t(X,y) = r(X,1) * (r(X,y) + s(X,y))
// (src/Sonic/Constraints.hs:61, where poly-0.4.0.0 runs a sparse convolution whose coefficients
// are themselves polynomials; evaluating Y := y first is a ring homomorphism, so the univariate
// product is result-exact).  omega_n = 7^((r-1)/n); Fr has 2-adicity 32.
//
// Forward = decimation in frequency (natural in, bit-reversed out); inverse = decimation in time
// (bit-reversed in, natural out, scaled by 1/n): the product needs no permutation pass.
// Stages whose butterfly span fits a 2048-element tile (64 KB of the CU's 160 KB LDS) run fused in
// one kernel out of LDS; wider stages stream through HBM one stage per launch.
#include "internal.hpp"

namespace sonic {

static constexpr int TILE_LOG = 11;
static constexpr int TILE = 1 << TILE_LOG;

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
__global__ __launch_bounds__(256) void k_ntt_stage(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s, int tw_shift, int inverse) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long halfn = 1L << (log2n - 1);
  if (t >= halfn) return;
  long half = 1L << (log2n - 1 - s);
  long j = t & (half - 1);
  long i0 = ((t >> (log2n - 1 - s)) << (log2n - s)) + j;
  long i1 = i0 + half;
  Fr w = tw[(j << s) << tw_shift];
  Fr a = d[i0], b = d[i1];
  if (!inverse) { d[i0] = fp_add(a, b); d[i1] = fp_mul(fp_sub(a, b), w); }
  else { Fr bw = fp_mul(b, w); d[i0] = fp_add(a, bw); d[i1] = fp_sub(a, bw); }
}

// all stages with span <= tile, fused in LDS.  tile_log = min(log2n, TILE_LOG).
__global__ __launch_bounds__(256) void k_ntt_local(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                                   int inverse, const Fr* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Fr* sh = reinterpret_cast<Fr*>(smem);
  const long base = (long)blockIdx.x << tile_log;
  const int tile = 1 << tile_log;
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
}

__global__ __launch_bounds__(256) void k_fr_scale(Fr* __restrict__ a, long n, const Fr* __restrict__ s) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = fp_mul(a[i], *s);
}
void fr_scale_enqueue(hipStream_t st, Fr* a, long n, const Fr* s) { LAUNCH(k_fr_scale, ceil_div(n, 256), 256, 0, st, a, n, s); }

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
    for (int s = 0; s < nglobal; s++) LAUNCH(k_ntt_stage, ceil_div(n / 2, 256), 256, 0, st, d, table, log2n, s, tw_shift, 0);
    LAUNCH(k_ntt_local, (int)(n >> tile_log), 256, lds, st, d, table, log2n, tile_log, tw_shift, 0, (const Fr*)nullptr);
  } else {
    const Fr* ninv = tw.ninv.as<Fr>() + log2n;
    if (nglobal == 0) {
      LAUNCH(k_ntt_local, (int)(n >> tile_log), 256, lds, st, d, table, log2n, tile_log, tw_shift, 1, ninv);
    } else {
      LAUNCH(k_ntt_local, (int)(n >> tile_log), 256, lds, st, d, table, log2n, tile_log, tw_shift, 1, (const Fr*)nullptr);
      for (int s = nglobal - 1; s >= 0; s--) LAUNCH(k_ntt_stage, ceil_div(n / 2, 256), 256, 0, st, d, table, log2n, s, tw_shift, 1);
      fr_scale_enqueue(st, d, n, ninv);
    }
  }
}

void ntt_forward_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, false); }
void ntt_inverse_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, true); }

__global__ __launch_bounds__(256) void k_fr_pointwise_mul(Fr* __restrict__ a, const Fr* __restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = fp_mul(a[i], b[i]);
}
void fr_pointwise_mul_enqueue(hipStream_t st, Fr* a, const Fr* b, long n) { LAUNCH(k_fr_pointwise_mul, ceil_div(n, 256), 256, 0, st, a, b, n); }

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
