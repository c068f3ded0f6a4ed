// Laurent-polynomial kernels of the prover, on dense Fr coefficient arrays (Montgomery form) over
// an exponent range [lo, lo + n).  The reference works on sparse (exponent, coeff) lists
// (poly-0.4.0.0); every operation here is result-exact, zeros simply stay in the array.
//
//   scale by powers     c_e <- c_e x^e        evalY on the diagonal r(X,Y) (src/Sonic/Utils.hs:20-21
//                                              with Constraints.hs:23-31), and the first half of eval
//   prefix sums         P_j = sum_{e<=j} c_e z^e   -> f(z) = P_hi         (`eval`, CommitmentScheme.hs:43)
//   quotient            w_j = z^{-1-j} (f(z) - P_j)  (j >= 0),  -z^{-1-j} P_j  (j < 0)
//                       = (f(X) - f(z)) / (X - z)                          (`divide`, CommitmentScheme.hs:44)
//   s(X,y), s(u,Y)      evalY / evalX of sPoly (Constraints.hs:34-53, Utils.hs:17-21)
//
// Quotient identity: for e > 0, (X^e - z^e)/(X - z) = sum_{k<e} X^k z^{e-1-k}; for e < 0,
// (X^e - z^e)/(X - z) = -sum_{m=1..|e|} X^{-m} z^{e+m-1}.  Collecting the coefficient of X^j gives
// the two prefix-sum forms above, so openPoly is one elementwise pass, one additive scan and one
// more elementwise pass -- no sequential Horner chain.
#include "internal.hpp"
#include "poly.hpp"

namespace sonic {

// Elements per thread of k_scale_powers.  A thread pays ~45 products for its starting power x^(e0 + i) and x^256 and then 2 per
// element, i.e. (45 + 2 PER) / PER products per element: 7.6 at PER = 8, 3.4 at 32.  These kernels touch ~77 n elements per proof
// (two passes per opening), so at 8 they were ~5 % of a proof's instruction count -- which is what streamed proofs pay for.
#ifndef SONIC_SCALE_PER
#define SONIC_SCALE_PER 32
#endif
constexpr int SCALE_PER = SONIC_SCALE_PER;
// ... for arrays long enough to fill the chip with threads of that length.  A short array (n = 2^14: 49 K coefficients = 6 workgroups at
// 32 per thread) runs as ONE dependent chain of 45 + 2 PER products on a few waves -- 80 us per launch, twenty launches on a proof's
// critical path (profiles/r06_small_proofs.txt) -- so the elements per thread shrink until ~64 workgroups exist (never below 2).  Not more
// workgroups than that: beside another proof's bucket accumulation (streamed proofs) wave slots come free at a few hundred workgroups
// per millisecond, and a launch of 257 short workgroups took 1-2 ms to be handed all of them (profiles/r06_small_proofs.txt).
static int scale_per(long n) {
  long per = n / (256L * 64L);
  if (per > SCALE_PER) per = SCALE_PER;
  if (per < 2) per = 2;
  return (int)per;
}

// out[i] = v(i) * x^(e0 + i); v(i) depends on mode:
//   0: in[i]                     1: 1 (pure power table)
//   2: quotient numerator: j = lo_q + i >= 0 ? F - P[i] : -P[i], with F = in[nF-1] (the last prefix)
template <int MODE>
__global__ __launch_bounds__(256) void k_scale_powers(const Fr* __restrict__ in, Fr* __restrict__ out, long n, long e0,
                                                      const Fr* __restrict__ px, const Fr* __restrict__ pxinv, long lo_q, long nF, int PER) {
  // PER elements per thread, strided by the block size
  const long base = (long)blockIdx.x * (256 * PER) + threadIdx.x;
  if (base >= n) return;
  const Fr x = *px, xinv = *pxinv;
  const long e = e0 + base;
  Fr p = e >= 0 ? fp_pow_u64(x, (uint64_t)e) : fp_pow_u64(xinv, (uint64_t)(-e));
  const Fr step = fp_pow_u64(x, 256);
  Fr F;
  if (MODE == 2) F = in[nF - 1];
#pragma unroll 1
  for (int k = 0; k < PER; k++) {
    const long i = base + (long)k * 256;
    if (i >= n) break;
    Fr v;
    if (MODE == 0) v = fp_mul(in[i], p);
    else if (MODE == 1) v = p;
    else { Fr P = in[i]; v = fp_mul((lo_q + i >= 0) ? fp_sub(F, P) : fp_neg(P), p); }
    out[i] = v;
    p = fp_mul(p, step);
  }
}

void poly_scale_powers_enqueue(hipStream_t st, const Fr* in, Fr* out, long n, long e0, const Fr* d_x, const Fr* d_xinv) {
  if (n <= 0) return;
  const int per = scale_per(n);
  if (in) LAUNCH(k_scale_powers<0>, ceil_div(n, 256L * per), 256, 0, st, in, out, n, e0, d_x, d_xinv, 0L, 0L, per);
  else LAUNCH(k_scale_powers<1>, ceil_div(n, 256L * per), 256, 0, st, in, out, n, e0, d_x, d_xinv, 0L, 0L, per);
}

// q[i] (exponent lo + i, i < n - 1) from the prefix sums P (n entries, exponents lo .. lo + n - 1)
void poly_quotient_enqueue(hipStream_t st, const Fr* prefix, Fr* q, long n, long lo, const Fr* d_z, const Fr* d_zinv) {
  if (n <= 1) return;
  // z^{-1-j} = (z^-1)^{1+j}: base z^-1 (inverse base z), first exponent 1 + lo
  const int per = scale_per(n - 1);
  LAUNCH(k_scale_powers<2>, ceil_div(n - 1, 256L * per), 256, 0, st, prefix, q, n - 1, 1 + lo, d_zinv, d_z, lo, n, per);
}

// Workgroups of an elementwise launch over `items` elements (256 threads each, grid-stride loops): short arrays get at most 64 fat
// workgroups (see scale_per: a launch of many short workgroups waits for wave slots one by one beside a running accumulation); long ones
// one element per thread, as before
static unsigned elementwise_grid(long items) {
  const long full = ceil_div(items, 256L);
  return (unsigned)(items <= (1L << 19) && full > 64 ? 64 : (full > 0 ? full : 1));
}

// ---- inclusive prefix sums in Fr (tile = 1024) -----------------------------------------------
__device__ __forceinline__ Fr block_inclusive_scan_fr(Fr v, Fr* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    Fr x = t >= o ? sh[t - o] : Fr::zero();
    __syncthreads();
    sh[t] = fp_add(sh[t], x);
    __syncthreads();
  }
  Fr r = sh[t];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void k_prefix_tiles(Fr* __restrict__ d, long n, Fr* __restrict__ tile_sums) {
  __shared__ Fr sh[256];
  const long base = (long)blockIdx.x * 1024 + threadIdx.x * 4;
  Fr v[4], s = Fr::zero();
  for (int k = 0; k < 4; k++) { v[k] = base + k < n ? d[base + k] : Fr::zero(); s = fp_add(s, v[k]); v[k] = s; }
  Fr incl = block_inclusive_scan_fr(s, sh);
  Fr excl = fp_sub(incl, s);
  for (int k = 0; k < 4; k++) if (base + k < n) d[base + k] = fp_add(v[k], excl);
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = incl;
}
__global__ __launch_bounds__(256) void k_prefix_top(Fr* __restrict__ tile_sums, long ntiles) {
  __shared__ Fr sh[256];
  __shared__ Fr carry_sh;
  if (threadIdx.x == 0) carry_sh = Fr::zero();
  __syncthreads();
  for (long base = 0; base < ntiles; base += 256) {
    long idx = base + threadIdx.x;
    Fr v = idx < ntiles ? tile_sums[idx] : Fr::zero();
    Fr incl = block_inclusive_scan_fr(v, sh);
    Fr carry = carry_sh;
    if (idx < ntiles) tile_sums[idx] = fp_add(fp_sub(incl, v), carry);   // exclusive
    __syncthreads();
    if (threadIdx.x == 255) carry_sh = fp_add(carry, incl);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_prefix_apply(Fr* __restrict__ d, long n, const Fr* __restrict__ tile_sums) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long tile = i >> 10;
  if (tile == 0) return;
  d[i] = fp_add(d[i], tile_sums[tile]);
}
void poly_prefix_sum_enqueue(hipStream_t st, Fr* d, long n, DevBuf& tmp) {
  if (n <= 0) return;
  const long ntiles = (n + 1023) / 1024;
  tmp.ensure(sizeof(Fr) * (ntiles + 1));
  LAUNCH(k_prefix_tiles, (int)ntiles, 256, 0, st, d, n, tmp.as<Fr>());
  if (ntiles > 1) {
    LAUNCH(k_prefix_top, 1, 256, 0, st, tmp.as<Fr>(), ntiles);
    LAUNCH(k_prefix_apply, ceil_div(n, 256), 256, 0, st, d, n, (const Fr*)tmp.as<Fr>());
  }
}

// ---- the same three steps for several openings at once (blockIdx.y = opening) -----------------
// Prefix sums over tiles of 256 * EPT coefficients (EPT = 4 or 16: a tile per workgroup, fewer and fatter workgroups for long arrays -- see
// scale_per); the tile offsets are added by the quotient kernel as it reads the prefixes, so there is no separate "apply" pass.
template <int MODE>
__global__ __launch_bounds__(256) void k_scale_powers_b(const OpenBatch b, long n, long e0, long lo_q, long nF, int PER, int tile_log) {
  const int y = blockIdx.y;
  const long base = (long)blockIdx.x * (256 * PER) + threadIdx.x;
  // MODE 0: D = poly * z^(e0 + i);  MODE 2: q = numerator(P) * (z^-1)^(e0 + i), P = D + its tile's offset (the prefix sums), F = P[nF - 1]
  const Fr* __restrict__ in = MODE == 0 ? b.poly[y] : b.D[y];
  const Fr* __restrict__ tiles = b.tiles[y];
  Fr* __restrict__ out = MODE == 0 ? b.D[y] : b.q[y];
  Fr F;
  if (MODE == 2) {
    F = in[nF - 1];
    if (((nF - 1) >> tile_log) > 0) F = fp_add(F, tiles[(nF - 1) >> tile_log]);
    if (blockIdx.x == 0 && threadIdx.x == 0) *b.fz[y] = F;      // f(z): the last prefix
  }
  if (base >= n) return;
  const Fr x = MODE == 0 ? b.zpair[y][0] : b.zpair[y][1], xinv = MODE == 0 ? b.zpair[y][1] : b.zpair[y][0];
  const long e = e0 + base;
  Fr p = e >= 0 ? fp_pow_u64(x, (uint64_t)e) : fp_pow_u64(xinv, (uint64_t)(-e));
  const Fr step = fp_pow_u64(x, 256);
#pragma unroll 1
  for (int k = 0; k < PER; k++) {
    const long i = base + (long)k * 256;
    if (i >= n) break;
    Fr v;
    if (MODE == 0) v = fp_mul(in[i], p);
    else {
      Fr P = in[i];
      if ((i >> tile_log) > 0) P = fp_add(P, tiles[i >> tile_log]);
      v = fp_mul((lo_q + i >= 0) ? fp_sub(F, P) : fp_neg(P), p);
    }
    out[i] = v;
    p = fp_mul(p, step);
  }
}
template <int EPT>
__global__ __launch_bounds__(256) void k_prefix_tiles_b(const OpenBatch b, long n) {
  __shared__ Fr sh[256];
  Fr* __restrict__ d = b.D[blockIdx.y];
  const long base = (long)blockIdx.x * (256 * EPT) + threadIdx.x * EPT;
  Fr s = Fr::zero();
  // (two passes over the thread's EPT consecutive elements instead of EPT live values: 16 x 8 registers would not leave room for the scan)
  for (int k = 0; k < EPT; k++) if (base + k < n) s = fp_add(s, d[base + k]);
  Fr incl = block_inclusive_scan_fr(s, sh);
  Fr run = fp_sub(incl, s);
  for (int k = 0; k < EPT; k++) if (base + k < n) { run = fp_add(run, d[base + k]); d[base + k] = run; }
  if (threadIdx.x == 255) b.tiles[blockIdx.y][blockIdx.x] = incl;
}
__global__ __launch_bounds__(256) void k_prefix_top_b(const OpenBatch b, long ntiles) {
  __shared__ Fr sh[256];
  __shared__ Fr carry_sh;
  Fr* __restrict__ tile_sums = b.tiles[blockIdx.x];
  if (threadIdx.x == 0) carry_sh = Fr::zero();
  __syncthreads();
  for (long base = 0; base < ntiles; base += 256) {
    long idx = base + threadIdx.x;
    Fr v = idx < ntiles ? tile_sums[idx] : Fr::zero();
    Fr incl = block_inclusive_scan_fr(v, sh);
    Fr carry = carry_sh;
    if (idx < ntiles) tile_sums[idx] = fp_add(fp_sub(incl, v), carry);   // exclusive
    __syncthreads();
    if (threadIdx.x == 255) carry_sh = fp_add(carry, incl);
    __syncthreads();
  }
}
// fz_k = the last prefix, for a batch whose quotients are not wanted (an evaluation only)
__global__ void k_last_prefix_b(const OpenBatch b, long len, int tile_log) {
  const int y = blockIdx.x;
  Fr F = b.D[y][len - 1];
  if (((len - 1) >> tile_log) > 0) F = fp_add(F, b.tiles[y][(len - 1) >> tile_log]);
  *b.fz[y] = F;
}
void open_batch_enqueue(hipStream_t st, const OpenBatch& b, long lo, long len, bool quotient) {
  if (b.k <= 0 || len <= 0) return;
  const unsigned k = (unsigned)b.k;
  const int tile_log = len >= 65536 ? 12 : 10;           // 4096- or 1024-coefficient tiles (OpenBatch::tiles holds len / 1024 + 2 entries)
  const int per = scale_per(len);
  LAUNCH(k_scale_powers_b<0>, dim3((unsigned)ceil_div(len, 256L * per), k), 256, 0, st, b, len, lo, 0L, 0L, per, tile_log);
  const long ntiles = (len + (1L << tile_log) - 1) >> tile_log;
  if (tile_log == 12) LAUNCH(k_prefix_tiles_b<16>, dim3((unsigned)ntiles, k), 256, 0, st, b, len);
  else LAUNCH(k_prefix_tiles_b<4>, dim3((unsigned)ntiles, k), 256, 0, st, b, len);
  if (ntiles > 1) LAUNCH(k_prefix_top_b, k, 256, 0, st, b, ntiles);
  if (!quotient) { LAUNCH(k_last_prefix_b, k, 1, 0, st, b, len, tile_log); return; }
  // (quotient exponents [lo, lo + len - 2]: z^{-1-j} = (z^-1)^{1 + j}; the launch also writes f(z) = the last prefix.  A polynomial of
  // ONE coefficient has an empty quotient: the launch then only writes f(z))
  const long qn = len > 1 ? len - 1 : 1;
  const int per2 = scale_per(qn);
  LAUNCH(k_scale_powers_b<2>, dim3((unsigned)ceil_div(qn, 256L * per2), k), 256, 0, st, b, len - 1 > 0 ? len - 1 : 0, 1 + lo, lo, len, per2, tile_log);
}

// ---- prover-specific builders ----------------------------------------------------------------
// r'(X,1) over [-2n-4, n]: aL at X^i, aR at X^-i, aO at X^{-i-n}, c_{n+i} at X^{-2n-i}
// (Constraints.hs:23-31, Protocol.hs:58-62).  Inputs Montgomery.
__global__ __launch_bounds__(256) void k_build_r1(const Fr* __restrict__ aL, const Fr* __restrict__ aR, const Fr* __restrict__ aO,
                                                  const Fr* __restrict__ cns, long n, Fr* __restrict__ r1) {
  const long len = 3 * n + 5;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) {   // index into [-2n-4, n]
    const long e = i - (2 * n + 4);
    Fr v = Fr::zero();
    if (e > 0) v = aL[e - 1];
    else if (e < 0 && e >= -n) v = aR[-e - 1];
    else if (e < -n && e >= -2 * n) v = aO[-e - n - 1];
    else if (e < -2 * n) v = cns[-e - 2 * n - 1];
    r1[i] = v;
  }
}
void build_r1_enqueue(hipStream_t st, const Fr* aL, const Fr* aR, const Fr* aO, const Fr* cns, long n, Fr* r1) {
  LAUNCH(k_build_r1, elementwise_grid(3 * n + 5), 256, 0, st, aL, aR, aO, cns, n, r1);
}

// s(X,y) over [-n, 2n] given ypow[e + n] = y^e for e in [-n, n + Q]:
//   X^-i: sum_q wL[q][i] y^{n+q};  X^i: sum_q wR[q][i] y^{n+q};
//   X^{i+n}: -y^i - y^-i + sum_q wO[q][i] y^{n+q}        (Constraints.hs:39-49, q = 1..Q)
__global__ __launch_bounds__(256) void k_s_of_y(const Fr* __restrict__ wL, const Fr* __restrict__ wR, const Fr* __restrict__ wO,
                                                const Fr* __restrict__ ypow, long n, long Q, Fr* __restrict__ s) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x + 1; i <= n + 1; i += (long)gridDim.x * 256) {   // 1..n
    if (i > n) { s[n] = Fr::zero(); break; }                 // X^0 slot
    Fr u = Fr::zero(), v = Fr::zero(), w = Fr::zero();
    for (long q = 0; q < Q; q++) {
      const Fr yq = ypow[2 * n + 1 + q];                     // y^{n+q+1}
      u = fp_add(u, fp_mul(wL[q * n + i - 1], yq));
      v = fp_add(v, fp_mul(wR[q * n + i - 1], yq));
      w = fp_add(w, fp_mul(wO[q * n + i - 1], yq));
    }
    w = fp_sub(fp_sub(w, ypow[n + i]), ypow[n - i]);
    s[n - i] = u;
    s[n + i] = v;
    s[2 * n + i] = w;
  }
}
void s_of_y_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, const Fr* ypow, long n, long Q, Fr* s) {
  LAUNCH(k_s_of_y, elementwise_grid(n + 1), 256, 0, st, wL, wR, wO, ypow, n, Q, s);
}

// s(X,Y) = sum_q Y^{n+q} P_q(X) + sum_i (-Y^i - Y^{-i}) X^{i+n}  with  P_q(X) = sum_i wL[q][i] X^{-i} + wR[q][i] X^i + wO[q][i] X^{i+n}
// (Constraints.hs:34-53, regrouped by constraint).  P_q depends on the circuit only, so Commit(P_q) is computed once per
// prover handle and Commit(s(X,y)) = sum_q y^{n+q} Commit(P_q) + Commit(diagonal part): an n-term MSM instead of a 3n-term one.
__global__ __launch_bounds__(256) void k_weight_row_poly(const Fr* __restrict__ wL, const Fr* __restrict__ wR, const Fr* __restrict__ wO,
                                                         long n, long q, Fr* __restrict__ s) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x + 1;   // 1..n
  if (i > n) { if (i == n + 1) s[n] = Fr::zero(); return; }
  s[n - i] = wL[q * n + i - 1];
  s[n + i] = wR[q * n + i - 1];
  s[2 * n + i] = wO[q * n + i - 1];
}
void weight_row_poly_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, long n, long q, Fr* s) {
  LAUNCH(k_weight_row_poly, ceil_div(n + 1, 256), 256, 0, st, wL, wR, wO, n, q, s);
}
// diag[i-1] = -(y^i + y^-i), i = 1..n (coefficients of X^{n+1..2n});  yq[q] = y^{n+1+q}
__global__ __launch_bounds__(256) void k_s_diag_part(const Fr* __restrict__ ypow, long n, long Q, Fr* __restrict__ diag, Fr* __restrict__ yq) {
  const long m = n > Q ? n : Q;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < m; t += (long)gridDim.x * 256) {
    if (t < n) diag[t] = fp_neg(fp_add(ypow[n + t + 1], ypow[n - t - 1]));
    if (t < Q) yq[t] = ypow[2 * n + 1 + t];
  }
}
void s_diag_part_enqueue(hipStream_t st, const Fr* ypow, long n, long Q, Fr* diag, Fr* yq) {
  LAUNCH(k_s_diag_part, elementwise_grid(n > Q ? n : Q), 256, 0, st, ypow, n, Q, diag, yq);
}

// s(u,Y) over [-n, n+Q] given upow[e + n] = u^e for e in [-n, 2n]:
//   Y^{+-i}: -u^{i+n};  Y^{n+q}: sum_i u^-i wL[q][i] + u^i wR[q][i] + u^{i+n} wO[q][i]   (Utils.hs:17-18)
__global__ __launch_bounds__(256) void k_s_of_u_diag(const Fr* __restrict__ upow, long n, Fr* __restrict__ s) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x + 1; i <= n + 1; i += (long)gridDim.x * 256) {
    if (i > n) { s[n] = Fr::zero(); break; }
    const Fr v = fp_neg(upow[2 * n + i]);
    s[n - i] = v;
    s[n + i] = v;
  }
}
// grid (blocks, Q): partial[q * nblk + blk] = sum over the block's i
__global__ __launch_bounds__(256) void k_s_of_u_rows(const Fr* __restrict__ wL, const Fr* __restrict__ wR, const Fr* __restrict__ wO,
                                                     const Fr* __restrict__ upow, long n, Fr* __restrict__ partial) {
  __shared__ Fr sh[256];
  const long q = blockIdx.y;
  Fr acc = Fr::zero();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x + 1; i <= n; i += (long)gridDim.x * 256) {
    acc = fp_add(acc, fp_mul(upow[n - i], wL[q * n + i - 1]));
    acc = fp_add(acc, fp_mul(upow[n + i], wR[q * n + i - 1]));
    acc = fp_add(acc, fp_mul(upow[2 * n + i], wO[q * n + i - 1]));
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fp_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[q * gridDim.x + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(64) void k_s_of_u_finish(const Fr* __restrict__ partial, int nblk, long n, long Q, Fr* __restrict__ s) {
  const long q = (long)blockIdx.x * 64 + threadIdx.x;
  if (q >= Q) return;
  Fr acc = Fr::zero();
  for (int b = 0; b < nblk; b++) acc = fp_add(acc, partial[q * nblk + b]);
  s[2 * n + 1 + q] = acc;
}
void s_of_u_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, const Fr* upow, long n, long Q, Fr* s, DevBuf& tmp) {
  LAUNCH(k_s_of_u_diag, elementwise_grid(n + 1), 256, 0, st, upow, n, s);
  int nblk = ceil_div(n, 256);
  if (nblk > 256) nblk = 256;
  if (n <= (1L << 17) && nblk > 64) nblk = 64;
  tmp.ensure(sizeof(Fr) * (size_t)nblk * Q);
  LAUNCH(k_s_of_u_rows, dim3(nblk, (unsigned)Q), 256, 0, st, wL, wR, wO, upow, n, tmp.as<Fr>());
  LAUNCH(k_s_of_u_finish, ceil_div(Q, 64), 64, 0, st, (const Fr*)tmp.as<Fr>(), nblk, n, Q, s);
}

// The two operands of t(X,y)'s product (Constraints.hs:56-65 with Y := y) in ONE launch: fa = r(X,1) and fb = r(X,y) + s(X,y), both over
// M transform points from exponent r_lo, zero beyond their coefficients.  (Two memsets of M elements, a copy, a scale-by-powers and an
// add-into before: five launches at the head of the longest dependent chain of a proof, each of which waits for wave slots when another
// proof's accumulation is running.)
__global__ __launch_bounds__(256) void k_t_operands(const Fr* __restrict__ r1, long r_len, long r_lo, const Fr* __restrict__ sy, long s_off, long s_len,
                                                    const Fr* __restrict__ ypair, Fr* __restrict__ fa, Fr* __restrict__ fb, long M, int PER) {
  const long base = (long)blockIdx.x * (256 * PER) + threadIdx.x;
  if (base >= M) return;
  Fr p = Fr::zero(), step = Fr::zero();
  if (base < r_len) {                                      // (r(X,y) = c_e y^e on the diagonal: Utils.hs:20-21)
    const Fr y = ypair[0], yinv = ypair[1];
    const long e = r_lo + base;
    p = e >= 0 ? fp_pow_u64(y, (uint64_t)e) : fp_pow_u64(yinv, (uint64_t)(-e));
    step = fp_pow_u64(y, 256);
  }
#pragma unroll 1
  for (int k = 0; k < PER; k++) {
    const long i = base + (long)k * 256;
    if (i >= M) break;
    Fr a = Fr::zero(), b = Fr::zero();
    if (i < r_len) { a = r1[i]; b = fp_mul(a, p); p = fp_mul(p, step); }
    if (i >= s_off && i < s_off + s_len) b = fp_add(b, sy[i - s_off]);
    fa[i] = a;
    fb[i] = b;
  }
}
void t_operands_enqueue(hipStream_t st, const Fr* r1, long r_len, long r_lo, const Fr* sy, long s_off, long s_len, const Fr* ypair, Fr* fa, Fr* fb, long M) {
  const int per = scale_per(M);
  LAUNCH(k_t_operands, ceil_div(M, 256L * per), 256, 0, st, r1, r_len, r_lo, sy, s_off, s_len, ypair, fa, fb, M, per);
}

// dst[off + i] += src[i]
__global__ __launch_bounds__(256) void k_add_into(Fr* __restrict__ dst, const Fr* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = fp_add(dst[i], src[i]);
}
void add_into_enqueue(hipStream_t st, Fr* dst, const Fr* src, long n) { if (n > 0) LAUNCH(k_add_into, elementwise_grid(n), 256, 0, st, dst, src, n); }

// *slot -= sum_q cs[q] * ypow_nq[q]   (k(y), Constraints.hs:67-68), then flags[flag_bit] if *slot != 0
__global__ __launch_bounds__(256) void k_sub_k_of_y(Fr* __restrict__ slot, const Fr* __restrict__ cs, const Fr* __restrict__ ypow_nq, long Q, int* flags, int flag_bit) {
  __shared__ Fr sh[256];
  Fr acc = Fr::zero();
  for (long q = threadIdx.x; q < Q; q += 256) acc = fp_add(acc, fp_mul(cs[q], ypow_nq[q]));
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fp_add(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Fr v = fp_sub(*slot, sh[0]);
    *slot = v;
    if (!v.is_zero()) atomicOr(flags, flag_bit);
  }
}
void sub_k_of_y_enqueue(hipStream_t st, Fr* slot, const Fr* cs, const Fr* ypow_nq, long Q, int* flags, int flag_bit) {
  LAUNCH(k_sub_k_of_y, 1, 256, 0, st, slot, cs, ypow_nq, Q, flags, flag_bit);
}

// flags |= bit if any of a[0..n) is non-zero
__global__ __launch_bounds__(256) void k_flag_nonzero(const Fr* __restrict__ a, long n, int* flags, int bit) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n && !a[i].is_zero()) atomicOr(flags, bit);
}
void flag_nonzero_enqueue(hipStream_t st, const Fr* a, long n, int* flags, int bit) {
  if (n > 0) LAUNCH(k_flag_nonzero, ceil_div(n, 256), 256, 0, st, a, n, flags, bit);
}

// ---- runs of equal coefficients (round 5) ---------------------------------------------------------------------------------------
// commitPoly of a coefficient vector that holds RUNS of one value c -- s(X, y_j) of a circuit whose weight rows repeat a value: n copies
// of two values among 3n + 1 coefficients for the reference's rndCircuit (test/Test/Reference.hs:141-155) -- needs only the two ends of a
// run: c (A[a] + ... + A[b]) = c ps[b] - c ps[a - 1] with the running sums ps of the basis (srs.hip).  Runs are found tile by tile
// (RUN_TILE consecutive coefficients all equal and non-zero); a tile's coefficients are zeroed in the copy the large MSM reads, and the
// tile contributes its closing term (+c, ps[last]) unless the next tile continues the run and its opening term (-c, ps[first - 1])
// unless the previous one does: 2 * ntiles slots, zero scalars where nothing is emitted, for one small MSM over gathered points.
__global__ __launch_bounds__(RUN_TILE) void k_run_tiles(const Fr* __restrict__ poly, long len, Fr* __restrict__ masked, Fr* __restrict__ val,
                                                       uint32_t* __restrict__ uniform, long ntiles) {
  __shared__ Fr first;
  const long t = blockIdx.x, i = t * RUN_TILE + threadIdx.x;
  const bool in = i < len;
  const Fr v = in ? poly[i] : Fr::zero();
  if (threadIdx.x == 0) first = v;
  __syncthreads();
  const Fr f = first;
  bool eq = in;
#pragma unroll
  for (int k = 0; k < 8; k++) eq = eq && v.l[k] == f.l[k];
  const bool uni = __syncthreads_and(eq ? 1 : 0) != 0 && !f.is_zero();
  if (in) masked[i] = uni ? Fr::zero() : v;
  if (threadIdx.x == 0 && t < ntiles) { uniform[t] = uni ? 1u : 0u; val[t] = f; }
}
__device__ __forceinline__ bool fr_same(const Fr& a, const Fr& b) {
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 8; k++) eq = eq && a.l[k] == b.l[k];
  return eq;
}
// ps: the running sums, positioned at the point of coefficient 0; ps_first: that point's index in the table (no ps[-1] at index 0)
__global__ __launch_bounds__(256) void k_run_terms(const Fr* __restrict__ val, const uint32_t* __restrict__ uniform, long ntiles, PointArray ps,
                                                   long ps_first, Fr* __restrict__ scal, G1Affine* __restrict__ pts) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  Fr s_close = Fr::zero(), s_open = Fr::zero();
  G1Affine p_close = G1Affine::inf(), p_open = G1Affine::inf();
  if (uniform[t]) {
    const Fr v = val[t];
    const bool cont_next = t + 1 < ntiles && uniform[t + 1] && fr_same(val[t + 1], v);
    const bool cont_prev = t > 0 && uniform[t - 1] && fr_same(val[t - 1], v);
    if (!cont_next) { s_close = v; p_close = ps[(size_t)((t + 1) * RUN_TILE - 1)]; }
    if (!cont_prev && ps_first + t * RUN_TILE > 0) { s_open = fp_neg(v); p_open = (ps + (t * RUN_TILE - 1))[0]; }
  }
  scal[2 * t] = s_close; scal[2 * t + 1] = s_open;
  pts[2 * t] = p_close; pts[2 * t + 1] = p_open;
}
void run_tiles_enqueue(hipStream_t st, const Fr* poly, long len, Fr* masked, Fr* val, uint32_t* uniform) {
  if (len <= 0) return;
  LAUNCH(k_run_tiles, ceil_div(len, (long)RUN_TILE), RUN_TILE, 0, st, poly, len, masked, val, uniform, len / RUN_TILE);
}
void run_terms_enqueue(hipStream_t st, const Fr* val, const uint32_t* uniform, long ntiles, PointArray ps, long ps_first, Fr* scal, G1Affine* pts) {
  if (ntiles > 0) LAUNCH(k_run_terms, ceil_div(ntiles, 256L), 256, 0, st, val, uniform, ntiles, ps, ps_first, scal, pts);
}

// small scalar prep: out = {v, v^-1} for each of k inputs (Montgomery in, Montgomery out); 0^-1 := 0
__global__ __launch_bounds__(64) void k_fr_with_inverse(const Fr* __restrict__ in, int k, Fr* __restrict__ out) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= k) return;
  Fr v = in[i];
  out[2 * i] = v;
  out[2 * i + 1] = v.is_zero() ? v : fp_inv(v);
}
void fr_with_inverse_enqueue(hipStream_t st, const Fr* in, int k, Fr* out) { LAUNCH(k_fr_with_inverse, ceil_div(k, 64), 64, 0, st, in, k, out); }

__global__ void k_fr_mul_scalar(const Fr* a, const Fr* b, Fr* out) { *out = fp_mul(*a, *b); }
void fr_mul_scalar_enqueue(hipStream_t st, const Fr* a, const Fr* b, Fr* out) { LAUNCH(k_fr_mul_scalar, 1, 1, 0, st, a, b, out); }

// sparse (sorted by exponent) -> dense: run heads sum their run
__global__ __launch_bounds__(256) void k_sparse_to_dense(const int64_t* __restrict__ exps, const Fr* __restrict__ coeffs, long nt, long lo, Fr* __restrict__ dense) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nt) return;
  if (i > 0 && exps[i - 1] == exps[i]) return;
  Fr acc = coeffs[i];
  for (long j = i + 1; j < nt && exps[j] == exps[i]; j++) acc = fp_add(acc, coeffs[j]);
  dense[exps[i] - lo] = acc;
}
void sparse_to_dense_enqueue(hipStream_t st, const int64_t* exps, const Fr* coeffs, long nt, long lo, Fr* dense) {
  if (nt > 0) LAUNCH(k_sparse_to_dense, ceil_div(nt, 256), 256, 0, st, exps, coeffs, nt, lo, dense);
}


// out[i] = c[i] * b^{e[i]} for the terms of a sparse polynomial (e may be negative: pair = {b, b^-1}): one variable of a sparse
// bivariate Laurent polynomial substituted (evalX / evalY, Utils.hs:17-21, on poly's sparse form)
__global__ __launch_bounds__(256) void k_scale_terms(const int64_t* __restrict__ e, const Fr* __restrict__ c, long nt, const Fr* __restrict__ pair, Fr* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nt) return;
  const int64_t ex = e[i];
  const Fr base = ex >= 0 ? pair[0] : pair[1];
  out[i] = fp_mul(c[i], fp_pow_u64(base, (uint64_t)(ex >= 0 ? ex : -ex)));
}
void scale_terms_enqueue(hipStream_t st, const int64_t* e, const Fr* c, long nt, const Fr* pair, Fr* out) {
  if (nt > 0) LAUNCH(k_scale_terms, ceil_div(nt, 256), 256, 0, st, e, c, nt, pair, out);
}

}  // namespace sonic
