// NTT over Fr (radix 2 butterflies in generated assembly, the wide stages five or six per memory pass) for the Laurent-polynomial product t(X,y) = r(X,1) * (r(X,y) + s(X,y))
// (src/Sonic/Constraints.hs:61, where poly-0.4.0.0 runs a sparse convolution whose coefficients
// are themselves polynomials; evaluating Y := y first is a ring homomorphism, so the univariate
// product is result-exact).  omega_n = 7^((r-1)/n); Fr has 2-adicity 32.
//
// Forward = decimation in frequency (natural in, bit-reversed out); inverse = decimation in time
// (bit-reversed in, natural out, scaled by 1/n): the product needs no permutation pass.
// Stages whose butterfly span fits a 2048-element tile (64 KB of the CU's 160 KB LDS) run fused in
// one kernel out of LDS (k_ntt_local); the wider ones run up to six at a time on blocks of rows x columns that
// pass through LDS once per launch (k_ntt_wide): three passes over HBM for a 2^21- or 2^23-point transform.
#include <algorithm>
#include <stdlib.h>
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

// Stage-major twiddles: stage s of a 2^L-point transform reads w^(j 2^s), j < 2^(L-1-s).  Out of the plain table tw[k] = w^k those are
// 2^s elements apart -- 32 KB between the twiddles of neighbouring lanes in the last wide stages: every lane its own cache line.
// st[off(s) + j] = w^(j 2^s) with off(s) = 2^L - 2^(L-s) puts every stage's twiddles side by side (twice the memory: 2^L entries).
// A 2^m-point transform inside a table built for 2^L reads stage s + (L - m): same layout, shifted.
__global__ __launch_bounds__(256) void k_ntt_stage_major(const Fr* __restrict__ tw, Fr* __restrict__ st, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;             // position in st: i = off(s) + j
  const long N = 1L << L;
  if (i >= N - 1) return;
  // s = number of leading ones of i read as an L-bit number: off(s) <= i < off(s + 1)
  int s = 0;
  while (i >= N - (N >> (s + 1))) s++;
  const long j = i - (N - (N >> s));
  st[i] = tw[j << s];
}
__device__ __forceinline__ long stage_off(int L, int s) { return (1L << L) - (1L << (L - s)); }

// The streaming kernels below run grid-stride over a capped grid (WIDE_GRID workgroups).  Launched beside a bucket accumulation,
// which holds every wave slot with long-lived waves, a kernel gets a slot only when an accumulation workgroup retires: with one
// short workgroup per 256 elements a 2^23-point stage needed 8192 such grants and took 36 ms instead of 0.15 ms; a few hundred
// long-lived workgroups need a few hundred.
static const int WIDE_GRID = getenv("SONIC_NTT_GRID") ? atoi(getenv("SONIC_NTT_GRID")) : 512;
// workgroup caps of the two transform kernels (tools/ntt_time.py: one block or tile per workgroup up to these caps measures best alone
// on the chip -- 0.444 against 0.465 ms for the LDS kernel at 1024 against 512 workgroups -- and the same inside prove)
static const int WIDE_BLOCKS = 4 * WIDE_GRID;      // 2048-element blocks (twice as many of 1024)
static const int LOCAL_BLOCKS = 2 * WIDE_GRID;     // 2048-element tiles
static inline int wide_grid(long items) { long g = (items + 255) / 256; return (int)(g < WIDE_GRID ? g : WIDE_GRID); }

// Several consecutive WIDE stages in one pass through HBM (round 4; before: two per pass as radix-4 butterflies in registers, 6 passes
// per 2^21-point transform, each bound by the ~4 TB/s its 32-byte strided accesses reach).  Stages s0 .. s0 + ns - 1 only connect
// elements whose indices differ in the ns bits below bit log2n - s0: a workgroup takes the 2^ns "rows" i = base + k * stride
// (stride = 2^(log2n - s0 - ns)) for C consecutive columns -- 1024 elements, 32 KB of LDS, C x 32 B contiguous per row -- runs the ns
// stages out of LDS and writes the block back in place.  Ten wide stages are two passes of five.
//
// The butterflies are the generated routines sonic_ntt_bfly2_fwd / _inv / _unit (mont_asm.hpp): the butterflies a thread owns in a
// stage as one scheduled program that reads its operands from LDS and its twiddles from the stage-major table, in the lazy range
// [0, 2r).  Two per thread keep a kernel within 128 VGPRs -- four waves per SIMD, which hide each other's LDS round trips and barriers
// (the four-per-thread routines, two waves per SIMD, are kept behind SONIC_NTT_WAVES=2: 7 % slower).
// Values are canonical again where they leave the transform: the forward transform's last store, the inverse transform's 1/n scaling.
static constexpr int WIDE_ELEMS_LOG = 11;      // elements per workgroup block (four butterflies per thread; 10 with two)

__device__ __forceinline__ uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)p; }      // the low half of a flat LDS pointer is the LDS offset
__device__ __forceinline__ Fr fr_canonical(const Fr& a) { return fp_add(a, Fr::zero()); }              // [0, 2r) -> [0, r)
// U butterflies of a thread in one stage (unit: the stage with a span of one element -- every twiddle is w^0 = 1, the butterfly is a
// sum and a difference).  U = 4: 176 VGPRs inside the routine, two waves per SIMD; U = 2: 106, four waves per SIMD.
// LDS layout of a block of E elements (round 5).  As 32-byte records (HI = 16: the upper four limbs right behind the lower four) every
// ds_read_b128 / ds_write_b128 of a wave touches every OTHER 16-byte slot -- a two-way bank conflict on all LDS traffic of the transform
// (rocprofv3 --pmc: SQ_LDS_BANK_CONFLICT = 63-68 % of SQ_LDS_IDX_ACTIVE, profiles/r05_lds_before.txt).  SPLIT: the lower halves of the
// block's elements form one plane (element i at byte 16 i), the upper halves a second one HI = 16 E bytes further; consecutive lanes then
// read consecutive slots.  HI is an immediate of the generated routines (sonic_ntt_bfly2s14_* for 1024-element blocks, ..s15_* for
// 2048-element tiles); a 4096-element block (HI = 65536 does not fit the 16-bit offset field) and the four-butterfly routines keep records.
template <int U, int ELOG> struct LdsLayout {
  static constexpr bool SPLIT = U == 2 && ELOG <= 11;
  static constexpr uint32_t STRIDE = SPLIT ? 16 : 32;
  static constexpr uint32_t HI = SPLIT ? (16u << ELOG) : 16u;
};
__device__ __forceinline__ void lds_put(unsigned char* sh, uint32_t stride, uint32_t hi, int i, const Fr& v) {
  const uint4* w = reinterpret_cast<const uint4*>(&v);
  *reinterpret_cast<uint4*>(sh + stride * (uint32_t)i) = w[0];
  *reinterpret_cast<uint4*>(sh + stride * (uint32_t)i + hi) = w[1];
}
__device__ __forceinline__ Fr lds_get(const unsigned char* sh, uint32_t stride, uint32_t hi, int i) {
  Fr v;
  uint4* w = reinterpret_cast<uint4*>(&v);
  w[0] = *reinterpret_cast<const uint4*>(sh + stride * (uint32_t)i);
  w[1] = *reinterpret_cast<const uint4*>(sh + stride * (uint32_t)i + hi);
  return v;
}

template <int U, uint32_t HI = 16>
__device__ __forceinline__ void ntt_bfly(int inverse, const uint32_t (&e0)[U], const uint32_t (&tj)[U], uint32_t span, const Fr* stw, bool unit = false) {
#if defined(__HIP_DEVICE_COMPILE__)       // (the generated routines exist in the device pass only)
#if defined(SONIC_NTT_PROBE)              // timing probes (tools only; results are wrong): 1 = no butterflies at all, 2 = every stage as the unit stage
  if (SONIC_NTT_PROBE == 1) return;
  unit = true;
#endif
  if constexpr (U == 4) {
    if (unit) sonic_ntt_bfly4_unit(e0[0], e0[1], e0[2], e0[3], span);
    else if (!inverse) sonic_ntt_bfly4_fwd(e0[0], e0[1], e0[2], e0[3], tj[0], tj[1], tj[2], tj[3], span, stw);
    else sonic_ntt_bfly4_inv(e0[0], e0[1], e0[2], e0[3], tj[0], tj[1], tj[2], tj[3], span, stw);
  } else if constexpr (HI == 16384) {
    if (unit) sonic_ntt_bfly2s14_unit(e0[0], e0[1], span);
    else if (!inverse) sonic_ntt_bfly2s14_fwd(e0[0], e0[1], tj[0], tj[1], span, stw);
    else sonic_ntt_bfly2s14_inv(e0[0], e0[1], tj[0], tj[1], span, stw);
  } else if constexpr (HI == 32768) {
    if (unit) sonic_ntt_bfly2s15_unit(e0[0], e0[1], span);
    else if (!inverse) sonic_ntt_bfly2s15_fwd(e0[0], e0[1], tj[0], tj[1], span, stw);
    else sonic_ntt_bfly2s15_inv(e0[0], e0[1], tj[0], tj[1], span, stw);
  } else {
    static_assert(HI == 16, "a split layout needs its own generated routines (tools/gen_mont_asm.py)");
    if (unit) sonic_ntt_bfly2_unit(e0[0], e0[1], span);
    else if (!inverse) sonic_ntt_bfly2_fwd(e0[0], e0[1], tj[0], tj[1], span, stw);
    else sonic_ntt_bfly2_inv(e0[0], e0[1], tj[0], tj[1], span, stw);
  }
#endif
}

// ELOG: log2 of the block's elements = 2 U x 256 threads
template <int U, int ELOG>
__device__ __forceinline__ void ntt_wide_body(Fr* sh_fr, Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s0, int ns, int tw_shift, int inverse,
                                              const Fr* __restrict__ scale) {
  constexpr int THREADS = (1 << ELOG) / (2 * U);            // U butterflies per thread and stage
  static_assert(THREADS == 256 || THREADS == 1024, "256 threads per block of 1024 / 2048 elements, 1024 for the 4096-element block");
  using LY = LdsLayout<U, ELOG>;
  unsigned char* sh = reinterpret_cast<unsigned char*>(sh_fr);
  const uint32_t lds0 = lds_address(sh);
  const int lc = ELOG - ns;                                 // log2 C
  const int lstride = log2n - s0 - ns;                      // log2 of the row stride
  const long col_blocks = 1L << (lstride - lc);
  const long nitems = col_blocks << s0;
  for (long item = blockIdx.x; item < nitems; item += gridDim.x) {
    const long u = item >> (lstride - lc), cb = item & (col_blocks - 1);
    const long base = (u << (log2n - s0)) + (cb << lc);
    for (int e = threadIdx.x; e < (1 << ELOG); e += THREADS) lds_put(sh, LY::STRIDE, LY::HI, e, d[base + ((long)(e >> lc) << lstride) + (e & ((1 << lc) - 1))]);
    __syncthreads();
    for (int t = 0; t < ns; t++) {
      const int tt = inverse ? ns - 1 - t : t;               // forward (DIF): widest span first; inverse (DIT): the reverse
      const int s = s0 + tt;
      const int hb = ns - 1 - tt;                            // the row bit this stage pairs
      const int lhalf = log2n - 1 - s;                       // log2 of the butterfly span in elements
      const Fr* stw = tw + stage_off(log2n + tw_shift, s + tw_shift);
      uint32_t e0[U], tj[U];
#pragma unroll
      for (int q = 0; q < U; q++) {
        const int bt = threadIdx.x + q * THREADS;
        const int c = bt & ((1 << lc) - 1), kp = bt >> lc;
        const int k = ((kp >> hb) << (hb + 1)) | (kp & ((1 << hb) - 1));         // row with bit hb clear
        const long j = (((long)(k & ((1 << hb) - 1)) << lstride) + (cb << lc) + c) & ((1L << lhalf) - 1);
        e0[q] = lds0 + (uint32_t)((k << lc) | c) * LY::STRIDE;
        tj[q] = (uint32_t)j * (uint32_t)sizeof(Fr);
      }
      const uint32_t span = LY::STRIDE << (hb + lc);
      ntt_bfly<U, LY::HI>(inverse, e0, tj, span, stw);
      __syncthreads();
    }
    if (scale) {
      const Fr sc = *scale;
      for (int e = threadIdx.x; e < (1 << ELOG); e += THREADS) d[base + ((long)(e >> lc) << lstride) + (e & ((1 << lc) - 1))] = fp_mul(lds_get(sh, LY::STRIDE, LY::HI, e), sc);
    } else {
      for (int e = threadIdx.x; e < (1 << ELOG); e += THREADS) d[base + ((long)(e >> lc) << lstride) + (e & ((1 << lc) - 1))] = lds_get(sh, LY::STRIDE, LY::HI, e);
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256, 2) void k_ntt_wide4(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s0, int ns, int tw_shift, int inverse,
                                                     const Fr* __restrict__ scale) {
  __shared__ __attribute__((aligned(16))) Fr sh[1 << WIDE_ELEMS_LOG];
  ntt_wide_body<4, WIDE_ELEMS_LOG>(sh, d, tw, log2n, s0, ns, tw_shift, inverse, scale);
}
// the default: 1024-element blocks, two butterflies per thread, four workgroups per CU (SONIC_NTT_WAVES=2 selects the kernel above)
__global__ __launch_bounds__(256, 4) void k_ntt_wide(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s0, int ns, int tw_shift, int inverse,
                                                      const Fr* __restrict__ scale) {
  __shared__ __attribute__((aligned(16))) Fr sh[1 << (WIDE_ELEMS_LOG - 1)];
  ntt_wide_body<2, WIDE_ELEMS_LOG - 1>(sh, d, tw, log2n, s0, ns, tw_shift, inverse, scale);
}

// Nine or ten wide stages in ONE pass (round 5; behind SONIC_NTT_BIG=1, see ntt_run): 4096-element blocks -- 1024 rows x 4 columns at ten stages (128-B runs), 512 x 8 at nine --
// in 128 KB of LDS, one 1024-thread workgroup per CU (16 waves: the same four per SIMD as k_ntt_wide), two butterflies per thread.  A
// transform of 2^20 or 2^21 points then makes TWO passes over HBM instead of three.  Measured (tools/ntt_time.py, product alone on the
// chip, same box, alternating): M = 2^21 0.891 / 0.895 -> 0.870 / 0.867 ms (wide part 0.466 -> 0.450), M = 2^20 0.474 -> 0.445 ms (wide
// part 0.244 -> 0.212); at M = 2^19 the 128 blocks leave half of the CUs idle (0.277 -> 0.313 ms) and eleven stages would mean 64-B
// runs, so the pass is used for nine and ten wide stages only.  The transforms stay bound by VALU issue (21 stages x 2^20 butterflies x
// ~356 instructions: 0.66 ms of pure issue per product at M = 2^21); what the fused pass saves is one exposed load / store phase.
// NOT the default: a workgroup that needs a whole CU (16 waves, 128 KB) starts late beside the bucket accumulation of another proof.
__global__ __launch_bounds__(1024, 1) void k_ntt_wide_big(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int s0, int ns, int tw_shift, int inverse,
                                                          const Fr* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
  ntt_wide_body<2, 12>(reinterpret_cast<Fr*>(smem_big), d, tw, log2n, s0, ns, tw_shift, inverse, scale);
}

// all stages with span <= tile, fused in LDS.  tile_log = min(log2n, TILE_LOG).  Full tiles (2048 elements: every transform the prover
// runs) go through the generated butterflies, U per thread and stage (1024 / U threads); smaller transforms through the plain C++ ones.
// FULL: only full tiles (the generated butterflies); !FULL: only the small transforms (plain C++ butterflies over 32-byte records) -- two
// kernels, so that the hot one does not carry the other's code and registers (with both in one kernel and the stage loop unrolled the split
// layout pushed k_ntt_local past 128 VGPRs: three waves per SIMD and 14 % slower, or ten spilled registers and 5 % more HBM traffic under a
// four-wave bound; now 127 VGPRs and no scratch)
template <int U, bool FULL>
__device__ __forceinline__ void ntt_local_body(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                               int inverse, const Fr* __restrict__ scale, const Fr* __restrict__ mul) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Fr* sh = reinterpret_cast<Fr*>(smem);
  const uint32_t lds0 = lds_address(sh);
  const int tile = 1 << tile_log;
  constexpr int THREADS = 1024 / U;
  using LY = LdsLayout<U, TILE_LOG>;       // full tiles (the generated butterflies); the small transforms of the C++ branch keep 32-byte records
  const long ntiles = 1L << (log2n - tile_log);
  for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {       // grid-stride over the tiles (see WIDE_GRID)
  const long base = tl << tile_log;
  // forward: stages s = log2n - tile_log .. log2n - 1 (half = tile/2 .. 1)
  // inverse: the same stages in reverse order (half = 1 .. tile/2)
  // (mul: the pointwise product of two transforms folded into the inverse transform's first load)
  if constexpr (FULL) {
    if (mul) {
      _Pragma("unroll 1")
      for (int i = threadIdx.x; i < (1 << TILE_LOG); i += THREADS) lds_put(smem, LY::STRIDE, LY::HI, i, fp_mul(d[base + i], mul[base + i])); }
    else {
      _Pragma("unroll 1")
      for (int i = threadIdx.x; i < (1 << TILE_LOG); i += THREADS) lds_put(smem, LY::STRIDE, LY::HI, i, d[base + i]); }
    __syncthreads();
    // (not unrolled: the compiler otherwise computes the eleven stages' addresses up front and keeps -- or spills -- them across the routine calls)
#pragma unroll 1
    for (int k = 0; k < TILE_LOG; k++) {
      const int hl = inverse ? k : TILE_LOG - 1 - k;       // log2(half)
      const int s = log2n - 1 - hl;
      const Fr* stw = tw + stage_off(log2n + tw_shift, s + tw_shift);
      uint32_t e0[U], tj[U];
#pragma unroll
      for (int q = 0; q < U; q++) {
        const int bt = threadIdx.x + q * THREADS;
        const int j = bt & ((1 << hl) - 1);
        e0[q] = lds0 + (uint32_t)(((bt >> hl) << (hl + 1)) + j) * LY::STRIDE;
        tj[q] = (uint32_t)(j * (int)sizeof(Fr));
      }
      const uint32_t span = LY::STRIDE << hl;
      ntt_bfly<U, LY::HI>(inverse, e0, tj, span, stw, hl == 0);
      __syncthreads();
    }
    // the lazy range ends here: canonical out of the forward transform (and out of an inverse one that is scaled here)
    if (scale) { Fr sc = *scale; _Pragma("unroll 1") for (int i = threadIdx.x; i < tile; i += THREADS) d[base + i] = fp_mul(lds_get(smem, LY::STRIDE, LY::HI, i), sc); }
    else if (!inverse) { _Pragma("unroll 1") for (int i = threadIdx.x; i < tile; i += THREADS) d[base + i] = fr_canonical(lds_get(smem, LY::STRIDE, LY::HI, i)); }
    else { _Pragma("unroll 1") for (int i = threadIdx.x; i < tile; i += THREADS) d[base + i] = lds_get(smem, LY::STRIDE, LY::HI, i); }
  } else {
    if (mul) { for (int i = threadIdx.x; i < tile; i += THREADS) sh[i] = fp_mul(d[base + i], mul[base + i]); }
    else { for (int i = threadIdx.x; i < tile; i += THREADS) sh[i] = d[base + i]; }
    __syncthreads();
    for (int k = 0; k < tile_log; k++) {
      const int hl = inverse ? k : tile_log - 1 - k;
      const int s = log2n - 1 - hl;
      for (int bt = threadIdx.x; bt < tile / 2; bt += THREADS) {
        const int j = bt & ((1 << hl) - 1);
        const int i0 = ((bt >> hl) << (hl + 1)) + j, i1 = i0 + (1 << hl);
        const Fr w = tw[stage_off(log2n + tw_shift, s + tw_shift) + j];
        const Fr a = sh[i0], b = sh[i1];
        if (!inverse) { sh[i0] = fp_add(a, b); sh[i1] = fp_mul(fp_sub(a, b), w); }
        else { const Fr bw = fp_mul(b, w); sh[i0] = fp_add(a, bw); sh[i1] = fp_sub(a, bw); }
      }
      __syncthreads();
    }
    if (scale) { Fr sc = *scale; for (int i = threadIdx.x; i < tile; i += THREADS) d[base + i] = fp_mul(sh[i], sc); }
    else { for (int i = threadIdx.x; i < tile; i += THREADS) d[base + i] = sh[i]; }
  }
  __syncthreads();
  }
}
__global__ __launch_bounds__(256, 2) void k_ntt_local4(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                                      int inverse, const Fr* __restrict__ scale, const Fr* __restrict__ mul) {
  ntt_local_body<4, true>(d, tw, log2n, tile_log, tw_shift, inverse, scale, mul);
}
__global__ __launch_bounds__(512, 4) void k_ntt_local(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                                       int inverse, const Fr* __restrict__ scale, const Fr* __restrict__ mul) {
  ntt_local_body<2, true>(d, tw, log2n, tile_log, tw_shift, inverse, scale, mul);
}
// transforms below 2^11 points (one tile of 2^tile_log elements per workgroup, plain C++ butterflies)
__global__ __launch_bounds__(512) void k_ntt_small(Fr* __restrict__ d, const Fr* __restrict__ tw, int log2n, int tile_log, int tw_shift,
                                                   int inverse, const Fr* __restrict__ scale, const Fr* __restrict__ mul) {
  ntt_local_body<2, false>(d, tw, log2n, tile_log, tw_shift, inverse, scale, mul);
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

// The butterfly routines address a stage's twiddles as a 64-bit SGPR base + a 32-bit UNSIGNED byte offset in a VGPR (global_load with
// saddr zero-extends the offset) + 16 for the second half: the first stage of a 2^L-point transform holds 2^(L-1) twiddles, the last of
// them at (2^(L-1) - 1) x 32 B, so L = 28 (largest offset 2^32 - 32) is the limit -- 8 GB of data, 16 GB of stage-major tables; the
// prover's n = 2^24 (7n + 9 coefficients) needs 2^27.  (Round 4 stopped one bit early, at 2^27.)
static constexpr int NTT_MAX_LOG2N = 28;
void NttTables::ensure(hipStream_t st, int need) {
  if (need > NTT_MAX_LOG2N) { set_error("NTT of 2^%d points: at most 2^%d are supported", need, NTT_MAX_LOG2N); throw HipFail{SONIC_ERR_INVALID_ARG}; }
  if (need <= log2n) return;
  HIP_OK(hipStreamSynchronize(st));   // earlier launches may still read the old tables
  long half = 1L << (need - 1);
  DevBuf plain(sizeof(Fr) * half);
  fwd.alloc(sizeof(Fr) * 2 * half);
  inv.alloc(sizeof(Fr) * 2 * half);
  for (int dir = 0; dir < 2; dir++) {
    LAUNCH(k_ntt_twiddles, ceil_div(ceil_div(half, 16), 256), 256, 0, st, plain.as<Fr>(), half, need, dir);
    LAUNCH(k_ntt_stage_major, ceil_div(2 * half, 256), 256, 0, st, (const Fr*)plain.as<Fr>(), (dir ? inv : fwd).as<Fr>(), need);
  }
  HIP_OK(hipStreamSynchronize(st));       // `plain` goes out of scope
  if (!ninv.p) { ninv.alloc(sizeof(Fr) * 33); LAUNCH(k_fr_inv_pow2, 1, 64, 0, st, ninv.as<Fr>()); }
  log2n = need;
}

// waves per SIMD the transform kernels are built for: 4 (two butterflies per thread, <= 128 VGPRs; default: 0.87 against 0.93 ms per product
// at M = 2^21, 3.20 against 3.48 at 2^23) or 2 (four butterflies per thread: k_ntt_wide4 / k_ntt_local4)
static const int NTT_WAVES = getenv("SONIC_NTT_WAVES") ? atoi(getenv("SONIC_NTT_WAVES")) : 4;
static void local_launch(hipStream_t st, int grid, size_t lds, Fr* d, const Fr* table, int log2n, int tile_log, int tw_shift, int inverse, const Fr* scale, const Fr* mul) {
  if (tile_log < TILE_LOG) LAUNCH(k_ntt_small, grid, 512, lds, st, d, table, log2n, tile_log, tw_shift, inverse, scale, mul);
  else if (NTT_WAVES == 4) LAUNCH(k_ntt_local, grid, 512, lds, st, d, table, log2n, tile_log, tw_shift, inverse, scale, mul);
  else LAUNCH(k_ntt_local4, grid, 256, lds, st, d, table, log2n, tile_log, tw_shift, inverse, scale, mul);
}
static void ntt_run(hipStream_t st, const NttTables& tw, Fr* d, int log2n, bool inverse, const Fr* mul = nullptr) {
  if (log2n == 0) return;
  const int tile_log = log2n < TILE_LOG ? log2n : TILE_LOG;
  const int tw_shift = tw.log2n - log2n;
  const Fr* table = (inverse ? tw.inv : tw.fwd).as<Fr>();
  const long n = 1L << log2n;
  const int nglobal = log2n - tile_log;
  const size_t lds = sizeof(Fr) << tile_log;
  // the wide stages in passes of at most WIDE_MAX_STAGES, as even as possible (10 -> 5 + 5, 12 -> 6 + 6, 13 -> 5 + 4 + 4)
  // nine or ten wide stages go in one pass through a 128-KB block (k_ntt_wide_big) where the device grants that much dynamic LDS
  // OFF by default: alone on the chip it is the faster kernel, inside streamed proofs its 1024-thread workgroup has to wait for a CU with no
  // accumulation wave left on it and the proofs get slower (same box, alternating, 20 streamed proofs at n = 2^18: 32.21 / 32.34 / 31.51 ms
  // per proof with it against 31.44 / 31.35 / 31.47 without; sonic_prove_batch over two handles 34.1-34.6 against 32.3-33.4).
  // SONIC_NTT_BIG=1 selects it (tests/test_gpu_configs.py holds it against the default kernels).
  static const bool big_on = getenv("SONIC_NTT_BIG") && atoi(getenv("SONIC_NTT_BIG")) == 1;
  bool big = big_on && (nglobal == 9 || nglobal == 10) && NTT_WAVES == 4;
  if (big) {
    DeviceCtx& dctx = current_ctx();
    if (dctx.ntt_big < 0) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_wide_big), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) (void)hipGetLastError();
      dctx.ntt_big = e == hipSuccess ? 1 : 0;
    }
    big = dctx.ntt_big == 1;
  }
  const int WIDE_MAX_STAGES = big ? nglobal : 6;
  const int passes = (nglobal + WIDE_MAX_STAGES - 1) / WIDE_MAX_STAGES;
  int first[8], count[8];
  for (int p = 0, s = 0; p < passes; p++) { count[p] = nglobal / passes + (p < nglobal % passes ? 1 : 0); first[p] = s; s += count[p]; }
  auto wide = [&](int p, const Fr* scale) {
    if (big) {
      LAUNCH(k_ntt_wide_big, (int)std::min<long>(1L << (log2n - 12), 512), 1024, 128 * 1024, st, d, table, log2n, first[p], count[p], tw_shift, inverse ? 1 : 0, scale);
      return;
    }
    const int elog = NTT_WAVES == 4 ? WIDE_ELEMS_LOG - 1 : WIDE_ELEMS_LOG;
    const long items = 1L << (log2n - elog);
    if (NTT_WAVES == 4) LAUNCH(k_ntt_wide, (int)std::min<long>(items, 2L * WIDE_BLOCKS), 256, 0, st, d, table, log2n, first[p], count[p], tw_shift, inverse ? 1 : 0, scale);
    else LAUNCH(k_ntt_wide4, (int)std::min<long>(items, WIDE_BLOCKS), 256, 0, st, d, table, log2n, first[p], count[p], tw_shift, inverse ? 1 : 0, scale);
  };
  if (!inverse) {
    for (int p = 0; p < passes; p++) wide(p, nullptr);
    local_launch(st, (int)std::min<long>(n >> tile_log, LOCAL_BLOCKS), lds, d, table, log2n, tile_log, tw_shift, 0, nullptr, nullptr);
  } else {
    const Fr* ninv = tw.ninv.as<Fr>() + log2n;
    local_launch(st, (int)std::min<long>(n >> tile_log, LOCAL_BLOCKS), lds, d, table, log2n, tile_log, tw_shift, 1, nglobal == 0 ? ninv : nullptr, mul);
    for (int p = passes - 1; p >= 0; p--) wide(p, p == 0 ? ninv : nullptr);      // the last pass also scales by 1/n
  }
}

void ntt_forward_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, false); }
void ntt_inverse_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n) { ntt_run(st, tw, d, log2n, true); }
// inverse transform of the pointwise product d[i] * other[i] (both in the bit-reversed order the forward transforms leave): the product
// is formed as the first stage loads its tile -- one pass over both arrays less than a separate pointwise kernel
void ntt_inverse_of_product_enqueue(hipStream_t st, const NttTables& tw, Fr* d, const Fr* other, int log2n) {
  if (log2n == 0) { fr_pointwise_mul_enqueue(st, d, other, 1); return; }
  ntt_run(st, tw, d, log2n, true, other);
}

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
