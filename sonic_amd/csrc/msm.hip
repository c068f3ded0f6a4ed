// G1 multi-scalar multiplication for gfx950: the kernel behind commitPoly / openPoly.
//
// Reference: the two folds  foldl' (\acc (e,v) -> acc <> (B[e] `mul` v)) mempty  at
// src/Sonic/CommitmentScheme.hs:25-29 (alpha basis) and :45-48 (plain basis): per-term 255-bit
// double-and-add.  Here: Pippenger's bucket method, laid out for one MI355X.
//
//   1. k_part_hist / scan / k_part_scatter   one thread per scalar: fold s into [0,(r-1)/2] (negating the point
//                      instead), signed base-2^c digits; entries split by the high bits of their bucket key into
//                      <= 1024-bucket partitions through per-workgroup LDS histograms
//   2. k_part_sort     one workgroup per partition: LDS count of the low key bits -> bucket offsets, final placement
//                      entries[off[bucket] ..] = point index | window | sign
//   3. k_border_*      counting sort of the buckets by size (largest first), so that equal-length walks share a wave
//   4. k_bucket_accum  one thread per bucket walks its entries: affine point gathered from HBM
//                      (96 B contiguous), XYZZ mixed addition.  Buckets far above the mean (the
//                      protocol produces them: s(X,y) has n equal coefficients when a weight row
//                      is all ones, test/Test/Reference.hs:143-145) are only recorded here; their entries,
//                      end to end, are cut into equal stretches, one per workgroup (k_heavy_accum, k_heavy_finish)
//   5. k_bucket_segments / k_group_sum / k_window_sum   sum_b b*B_b per bucket set by running sums over K-bucket
//                      segments, then LDS trees
//   6. host tail       Horner over the W window sums + normalisation to the canonical affine
//                      bytes (msm_finish_host): O(W) sequential work, deferred by the caller
//
// msm_enqueue_batch runs k MSMs that share a plan as ONE such chain (job j = bucket set j): see msm.hpp.
//
// Algorithmic traffic: 96 B point + 32 B scalar per term read once from HBM per window pass;
// the kernel is integer-issue bound (v_mad_u64_u32), not HBM bound -- see DESIGN.md.
#include <string.h>
#include <stdexcept>
#include "internal.hpp"
#include "g1_quad.hpp"
#include "endo.hpp"

namespace sonic {

#ifndef SONIC_PART_LOW_BITS
#define SONIC_PART_LOW_BITS 8
#endif
constexpr int PART_LOW_BITS = SONIC_PART_LOW_BITS;   // buckets per sort partition = 2^PART_LOW_BITS (8..10)
#ifndef SONIC_PART_TILE
#define SONIC_PART_TILE 1024
#endif
constexpr int PART_TILE = SONIC_PART_TILE;            // scalars per tile in pass 1
constexpr uint32_t PASS1_GRID = 1024;      // workgroups of pass 1 (grid-stride over the tiles)
// workgroups of pass 2 (grid-stride over the (job, partition) pairs): one MSM over 2^19 buckets has 2048 partitions and keeps one
// workgroup each; a batched group of three or four would otherwise queue 6144-8192 short workgroups behind the wave slots that other
// groups' accumulations hold
constexpr uint32_t PART_SORT_GRID = 2048;
#ifndef SONIC_PART_SORT_THREADS
#define SONIC_PART_SORT_THREADS 1024
#endif
constexpr int PART_SORT_THREADS = SONIC_PART_SORT_THREADS;
#ifndef SONIC_PART_SCATTER_THREADS
#define SONIC_PART_SCATTER_THREADS 512
#endif
// lanes of a staged pass-1 workgroup (PART_TILE scalars per tile: a multiple of it).  Its 148-KB stage allows one workgroup per CU: 8 waves
// instead of 4 to hide the loads and LDS atomics.  Measured (N = 2^20 MSM, ms in this kernel; n = 2^20 proofs streamed): 256 lanes 0.126, 117.6-118.1;
// 512 lanes 0.0955, 116.7-116.9; 1024 lanes 0.0865, 117.5 (and n = 2^18 proofs 32.9-33.2 against 32.2-32.7: a 16-wave workgroup waits longer
// for a CU beside the accumulations)
constexpr int PART_SCATTER_THREADS = SONIC_PART_SCATTER_THREADS;
static_assert(PART_TILE % PART_SCATTER_THREADS == 0 && PART_SCATTER_THREADS >= 256, "a tile is a whole number of rounds of the workgroup");   // lanes of a pass-2 workgroup (its LDS stage allows one workgroup per CU)
static_assert(PART_LOW_BITS >= 8 && PART_LOW_BITS <= 10, "k_part_sort scans 256 x {1, 2, 4} counters");

static int g_window_override = 0;
int msm_window_override() { return g_window_override; }
void msm_set_window_override(int c) { g_window_override = c; }

// Heavy buckets (the unprepared S_j of a circuit whose constraint rows repeat a value: 2n of its 3n + 1 scalars are two values, so two
// buckets per window hold n entries each -- two thirds of the MSM's additions).  k_bucket_accum only records them (HeavyRec); their
// entries, laid end to end, are then cut into HEAVY_GRID equal stretches, one per workgroup of HEAVY_THREADS lanes (k_heavy_accum: one
// workgroup per CU, two waves per SIMD -- the occupancy the registers allow anyway), so that every lane of the chip chains the same
// number of additions whatever the buckets' sizes are; a stretch that crosses a bucket border yields one partial sum per bucket, and
// k_heavy_finish adds a bucket's partials.  One addition on a wave takes 10-20 us: the trees at the end of a stretch run on quads
// (g1_quad.hpp), four lanes per addition.
// Usually there is nothing heavy and the launch only has to notice that; on a chip full of other MSMs' accumulation every
// workgroup still waits for a free slot (1024 idle workgroups delayed the chain behind them by 1.4-3.4 ms in prove()).
#ifndef SONIC_HEAVY_THREADS
#define SONIC_HEAVY_THREADS 512
#endif
#ifndef SONIC_HEAVY_GRID
#define SONIC_HEAVY_GRID 256
#endif
static constexpr int HEAVY_THREADS = SONIC_HEAVY_THREADS;
static constexpr int HEAVY_GRID = SONIC_HEAVY_GRID;
static constexpr uint32_t HEAVY_MIN_STRETCH = 4 * HEAVY_THREADS;

// header of a chain, cleared by one memset at its start: the heavy-bucket counter and the size-class histogram / cursors of the
// bucket ordering (k_part_sort, k_border_place)
struct HeavyMeta { uint32_t pad, n_heavy; uint32_t class_hist[256]; uint32_t class_cursor[256]; };
struct HeavyRec { uint32_t bucket, cnt, base, npart; };     // bucket, cnt: k_bucket_accum; first partial and their number: k_heavy_accum

static void plan_finish(MsmPlan& p, long n) {
  p.NB = 1 << (p.c - 1);
  p.K = p.NB < 8 ? p.NB : 8;     // 8: shortest dependent chain (16 additions + one small scalar multiplication) that still fits one wave per SIMD
  p.nseg = p.NB / p.K;
  long mean = n * (long)(p.W / p.Wb) / p.NB;
  long t = 8 * mean;
  if (t < 256) t = 256;
  p.heavy_threshold = (uint32_t)t;
}

void msm_plan_set_segment(MsmPlan& p, int K) {
  if (K > p.NB) K = p.NB;
  if (K < 1) K = 1;
  while (K & (K - 1)) K &= K - 1;      // k_bucket_segments needs a power of two (NB is one, so K then divides it): round down
  p.K = K;
  p.nseg = p.NB / K;
}

MsmPlan msm_plan(long n, bool fold) {
  MsmPlan p;
  p.fold = fold;
  int lg = 0;
  while ((1L << (lg + 1)) <= n) lg++;
  int c = lg - 4;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  if (g_window_override >= 4 && g_window_override <= 16) c = g_window_override;
  p.c = c;
  // folded scalars are < 2^254, so the signed recoding never carries out of ceil(255 / c) windows; unfolded ones (< r < 2^255)
  // need room for that carry
  p.W = ((fold ? 255 : 256) + c - 1) / c;
  p.Wb = p.W;
  p.table_stride = 0;
  plan_finish(p, n);
  return p;
}

// Fixed-base plan over precomputed window tables (tab[w][i] = 2^(c w) P_i): every window feeds ONE shared
// set of 2^(c-1) buckets, so there is one running-sum reduction instead of W and no Horner tail, and c can
// grow (fewer windows => fewer point additions) without multiplying the bucket count by W.
MsmPlan msm_plan_tables(long n, int c, int W, long table_stride, bool endo) {
  MsmPlan p;
  p.fold = true;
  p.endo = endo;
  p.bits = endo ? ENDO_BITS : 255;
  p.c = c;
  p.W = W;
  p.Wb = 1;
  p.table_stride = table_stride;
  plan_finish(p, n);
  return p;
}

void MsmWorkspace::reserve(long n, const MsmPlan& pl, int k) {
  const size_t sets = (size_t)k * pl.Wb;
  size_t M = sets * pl.NB;
  size_t NW = (size_t)n * pl.W;
  const size_t part_hn = (size_t)ceil_div((long)pl.Wb * pl.NB, 1L << PART_LOW_BITS) * (size_t)(ceil_div(n, PART_TILE) + k);   // partitions per job x pass-1 workgroups
  count.ensure((2 * part_hn + 4) * 4);
  off.ensure((M + 1) * 4);
  digits.ensure(NW * 8);
  entries.ensure(NW * 4);
  buckets.ensure(M * sizeof(G1XYZZ));
  segres.ensure((sets * pl.nseg + sets * (pl.nseg / 256 + 1) + 2) * sizeof(G1XYZZ));
  scan_tmp.ensure((part_hn / 2048 + 4) * 4);
  order.ensure((M + 1) * 4);
  size_t max_heavy = NW / pl.heavy_threshold + 1;
  heavy_meta.ensure(sizeof(HeavyMeta) + max_heavy * sizeof(HeavyRec));
  heavy_partial.ensure((max_heavy + HEAVY_GRID + 1) * sizeof(G1XYZZ));      // a stretch border inside a bucket adds one partial
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fr_gt_half(const Fr& s) {
  constexpr uint32_t half[8] = FR_HALF;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (s.l[i] > half[i]) return true;
    if (s.l[i] < half[i]) return false;
  }
  return false;
}

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* sh, uint32_t* total);
// size class of a bucket for the largest-first ordering (256 classes; everything of 255 entries and more is class 0)
__device__ __forceinline__ uint32_t size_class(uint32_t sz) { return 255u - (sz < 255u ? sz : 255u); }

// ---- digits + two-pass partition sort ---------------------------------------------------------
// Every (scalar, window) pair becomes an entry (point index | window | sign) that must end up grouped by its
// bucket key.  A global histogram with returning atomics plus a random scatter moved ~2 GB per 2^20-term MSM and
// was bound by L2 atomic throughput.  Instead: pass 1 splits the keys by their high bits into partitions of
// 2^PART_LOW_BITS consecutive buckets with per-workgroup LDS histograms (k_part_hist, scan, k_part_scatter); pass 2 gives
// one workgroup per partition, which counts the low bits in LDS, scans them into the bucket offsets and places the
// entries (k_part_sort).  Order inside a bucket is irrelevant (the sum is commutative), so nothing needs to be stable.
// Measured at N = 2^20, 2^19 buckets (scatter + sort, ms): 1024-bucket partitions 0.21 + 0.28, 512: 0.26 + 0.20, 256: 0.30 + 0.13;
// then, at 256: XCD-contiguous tiles (xcd_tile) 0.25 + 0.13, and one 8-B (key, payload) record per entry instead of a 2-B and
// a 4-B store 0.16 + 0.145.

// kernel-argument view of a batch (by value); tile0[j] = first pass-1 workgroup of job j
struct MsmBatchDev {
  int k;
  int bits;                              // what the even-width windows cover (255, or 130 for the halves of an endomorphism split)
  int slot_off[MSM_MAX_JOBS];            // first entry of MsmSlot::win the job's sums go to (0; MSM_ENDO_SLOT_OFFSET for a second half)
  int slot_form;                         // MsmSlot::pad1 of the jobs' slots: 1 bit sums, 2 endomorphism pair
  uint32_t tile0[MSM_MAX_JOBS + 1];
  const char* points[MSM_MAX_JOBS];
  uint32_t pt_stride;                    // bytes between the points of every job (PointArray)
  long tstride[MSM_MAX_JOBS];            // points between a job's window tables (0: no tables)
  const Fr* scalars[MSM_MAX_JOBS];
  long n[MSM_MAX_JOBS];
  MsmSlot* slot[MSM_MAX_JOBS];
};
// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.  Pass 1 appends to 2^11 partition cursors per
// job and consecutive tiles append to ADJACENT addresses of every partition, so with tile = blockIdx the 64-B lines under the
// cursors are shared by workgroups on different XCDs and leave their L2s as partial writes (measured: 0.9 GB of HBM writes for
// 82 MB of entries).  Giving XCD x the contiguous tile range [x T/8, (x+1) T/8) keeps every line inside one L2.
// The pass-1 kernels run grid-stride over a capped grid (long-lived workgroups are not starved of wave slots beside another
// MSM's accumulation, see ntt.hip): workgroup b, on XCD b & 7, walks its share of that XCD's contiguous tile range.
struct TileWalk { uint32_t cur, end, step; };
__device__ __forceinline__ TileWalk xcd_tile_walk(uint32_t b, uint32_t G, uint32_t T) {
  if (G < 8) return TileWalk{b, T, G};
  const uint32_t x = b & 7u, k = b >> 3, Gx = (G + 7u - x) >> 3, per = T >> 3, rem = T & 7u;
  const uint32_t start = x * per + (x < rem ? x : rem), cnt = per + (x < rem ? 1u : 0u);
  return TileWalk{start + k, start + cnt, Gx};
}

__device__ __forceinline__ int batch_job_of_tile(const MsmBatchDev& b, uint32_t tile) {
  int j = 0;
  while (j + 1 < b.k && tile >= b.tile0[j + 1]) j++;
  return j;
}

struct DigitStream {
  Fr s;
  bool neg;
  uint32_t carry;
  __device__ __forceinline__ void init(const Fr* __restrict__ sc, long i, bool live, int mont, int fold) {
    s = live ? sc[i] : Fr::zero();
    if (mont) s = fp_from_mont(s);
    neg = fold && fr_gt_half(s);
    if (neg) s = fp_neg(s);          // r - s, still standard form
    carry = 0;
  }
  // next signed base-2^c digit: magnitude (0 = skip) and sign
  __device__ __forceinline__ uint32_t next(int c, uint32_t& sign) {
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint32_t d = (s.l[0] & mask) + carry;
#pragma unroll
    for (int k = 0; k < 7; k++) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));   // static limb indices stay in registers
    s.l[7] >>= c;
    sign = neg ? 1u : 0u;
    if (d > half) { d = (1u << c) - d; carry = 1; sign ^= 1u; } else carry = 0;
    return d;
  }
};

// LDS counter increment that returns the old value, with the lanes that share a key served by ONE LDS atomic: runs of
// equal scalars are the norm in this protocol (s(X,y) carries n copies of one coefficient when a weight row is all
// ones, test/Test/Reference.hs:143-145) and would otherwise serialise a whole wave on one LDS word.
__device__ __forceinline__ uint32_t lds_take(uint32_t* cnt, bool valid, uint32_t idx) {
  const int lane = threadIdx.x & 63;
  uint32_t r = 0;
  bool pending = valid;
  for (int round = 0; round < 2; round++) {
    const unsigned long long act = __ballot(pending);
    if (__popcll(act) < 16) break;
    const int first_lane = __ffsll((long long)act) - 1;
    const uint32_t k0 = __shfl(idx, first_lane);
    const bool same = pending && idx == k0;
    const unsigned long long m = __ballot(same);
    if (__popcll(m) < 8) break;
    uint32_t base = 0;
    if (lane == first_lane) base = atomicAdd(&cnt[k0], (uint32_t)__popcll(m));
    base = __shfl(base, first_lane);
    if (same) { r = base + (uint32_t)__popcll(m & ((1ull << lane) - 1)); pending = false; }
  }
  if (pending) r = atomicAdd(&cnt[idx], 1u);
  return r;
}

// Histogram layout: job-major, then partition, then the job's workgroups -- index (j, t, b) = P tile0[j] + t nblk_j + b, P =
// partitions per job.  Its exclusive scan is the write cursor of every (partition, workgroup) pair.
__global__ __launch_bounds__(256) void k_part_hist(const MsmBatchDev batch, int c, int W, int keystride, int mont, int fold, int P,
                                                   uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t h[];
  for (TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, batch.tile0[batch.k]); tw.cur < tw.end; tw.cur += tw.step) {
  for (int t = threadIdx.x; t < P; t += 256) h[t] = 0;
  __syncthreads();
  const uint32_t tile = tw.cur;
  const int job = batch_job_of_tile(batch, tile);
  const uint32_t blk = tile - batch.tile0[job], nblk = batch.tile0[job + 1] - batch.tile0[job];
  const Fr* __restrict__ sc = batch.scalars[job];
  const long n = batch.n[job];
  for (int k = 0; k < PART_TILE / 256; k++) {
    const long i = (long)blk * PART_TILE + k * 256 + threadIdx.x;
    DigitStream ds;
    ds.init(sc, i, i < n, mont, fold);
    for (int w = 0; w < W; w++) {
      uint32_t sign;
      const uint32_t d = ds.next(keystride ? c : msm_even_width(W, w, batch.bits), sign);       // shared buckets over tables: even widths
      const uint32_t key = (uint32_t)w * keystride + d - 1;
      lds_take(h, d != 0, key >> PART_LOW_BITS);
    }
  }
  __syncthreads();
  uint32_t* out = hist + (size_t)P * batch.tile0[job] + blk;
  for (int t = threadIdx.x; t < P; t += 256) out[(size_t)t * nblk] = h[t];
  __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_part_scatter(const MsmBatchDev batch, int c, int W, int keystride, int mont, int fold, int P,
                                                      const uint32_t* __restrict__ base, uint2* __restrict__ part) {
  extern __shared__ uint32_t cur[];
  for (TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, batch.tile0[batch.k]); tw.cur < tw.end; tw.cur += tw.step) {
  const uint32_t tile = tw.cur;
  const int job = batch_job_of_tile(batch, tile);
  const uint32_t blk = tile - batch.tile0[job], nblk = batch.tile0[job + 1] - batch.tile0[job];
  const Fr* __restrict__ sc = batch.scalars[job];
  const long n = batch.n[job];
  const uint32_t* in = base + (size_t)P * batch.tile0[job] + blk;
  for (int t = threadIdx.x; t < P; t += 256) cur[t] = in[(size_t)t * nblk];
  __syncthreads();
  for (int k = 0; k < PART_TILE / 256; k++) {
    const long i = (long)blk * PART_TILE + k * 256 + threadIdx.x;
    DigitStream ds;
    ds.init(sc, i, i < n, mont, fold);
    for (int w = 0; w < W; w++) {
      uint32_t sign;
      const uint32_t d = ds.next(keystride ? c : msm_even_width(W, w, batch.bits), sign);       // shared buckets over tables: even widths
      const uint32_t key = (uint32_t)w * keystride + d - 1;
      const uint32_t pos = lds_take(cur, d != 0, key >> PART_LOW_BITS);
      if (d) {
        // one 8-B store per entry: (low key bits, payload = index | window (tables only) | sign)
        part[pos] = make_uint2(key & ((1u << PART_LOW_BITS) - 1), (uint32_t)i | (keystride ? 0u : (uint32_t)w << 26) | (sign << 31));
      }
    }
  }
  __syncthreads();
  }
}

// The same pass with its output STAGED in LDS (round 4).  The direct kernel above issues every record as its own 8-byte store to one of
// 2048 partition cursors: 64 partial lines per wave store, 440 MB written for 109 MB of records (profiles/r03_pmc_msm.json).  Here the
// workgroup first sorts its tile's records by partition inside LDS (the per-tile counts are the histogram's, so the local offsets are
// a 2048-entry scan), then copies the sorted buffer out: consecutive lanes hold consecutive records of a run, and a run goes to
// consecutive addresses.  LDS: records 8 B + partition id 2 B per (scalar, window) + two words per partition -- 148 KB at 13 windows
// and 2048 partitions, one workgroup per CU (the pass is short); plans that need more than SORT_STAGE_MAX_LDS use the direct kernel.
constexpr size_t SORT_STAGE_MAX_LDS = 156 * 1024;
// PLANES (round 6, VERDICT r05 item 6: the split-plane layout that removed the transforms' LDS bank conflicts, tried here): the two words of
// a record in two planes of 4-byte slots instead of one array of 8-byte records.  Measured -- profiles/r06_sort_planes_ab.txt -- and NOT the
// default: the records are scattered to RANDOM slots (that is what a counting sort's placement is), so a wave's 64 writes meet the 64 banks
// like 64 balls thrown at 64 bins whether a slot is one bank wide or two; the conflicts of this kernel and of k_part_sort (42-54 % of the
// LDS-active cycles) are those of the placement and of the LDS atomics on random counters, not of a layout.
template <bool PLANES>
__global__ __launch_bounds__(PART_SCATTER_THREADS) void k_part_scatter_staged(const MsmBatchDev batch, int c, int W, int keystride, int mont, int fold, int P,
                                                             const uint32_t* __restrict__ hist, const uint32_t* __restrict__ base,
                                                             uint2* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_stage[];
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem_stage);                 // [P]  running local cursor
  uint32_t* delta = cur + P;                                                // [P]  global position - local position of the partition's run
  uint2* recs = reinterpret_cast<uint2*>(delta + P);                        // [PART_TILE * W]
  uint32_t* rec_lo = reinterpret_cast<uint32_t*>(recs);                     // (PLANES: the same bytes as two arrays of words)
  uint32_t* rec_hi = rec_lo + (size_t)PART_TILE * W;
  uint16_t* pid = reinterpret_cast<uint16_t*>(recs + (size_t)PART_TILE * W);  // [PART_TILE * W]
  __shared__ uint32_t sc[256];
  __shared__ uint32_t total_sh;
  constexpr int T = PART_SCATTER_THREADS;       // (the first 256 threads scan the tile's partition counts; all of them place records)
  for (TileWalk tw = xcd_tile_walk(blockIdx.x, gridDim.x, batch.tile0[batch.k]); tw.cur < tw.end; tw.cur += tw.step) {
    const uint32_t tile = tw.cur;
    const int job = batch_job_of_tile(batch, tile);
    const uint32_t blk = tile - batch.tile0[job], nblk = batch.tile0[job + 1] - batch.tile0[job];
    const Fr* __restrict__ scal = batch.scalars[job];
    const long n = batch.n[job];
    const size_t row0 = (size_t)P * batch.tile0[job] + blk;
    // local exclusive scan of this tile's per-partition counts: ceil(P / 256) consecutive partitions per thread
    const int per = (P + 255) / 256;
    const bool scans = threadIdx.x < 256;
    uint32_t sum = 0;
    for (int k = 0; scans && k < per; k++) { const int t = threadIdx.x * per + k; if (t < P) sum += hist[row0 + (size_t)t * nblk]; }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan_256(sum, sc, &tot);
    for (int k = 0; scans && k < per; k++) {
      const int t = threadIdx.x * per + k;
      if (t < P) { cur[t] = ex; delta[t] = base[row0 + (size_t)t * nblk] - ex; ex += hist[row0 + (size_t)t * nblk]; }
    }
    if (threadIdx.x == 0) total_sh = tot;
    __syncthreads();
    for (int k = 0; k < PART_TILE / T; k++) {
      const long i = (long)blk * PART_TILE + k * T + threadIdx.x;
      DigitStream ds;
      ds.init(scal, i, i < n, mont, fold);
      for (int w = 0; w < W; w++) {
        uint32_t sign;
        const uint32_t d = ds.next(keystride ? c : msm_even_width(W, w, batch.bits), sign);
        const uint32_t key = (uint32_t)w * keystride + d - 1;
        const uint32_t pr = key >> PART_LOW_BITS;
        const uint32_t pos = lds_take(cur, d != 0, pr);
        if (d) {
          const uint32_t lo = key & ((1u << PART_LOW_BITS) - 1), hi = (uint32_t)i | (keystride ? 0u : (uint32_t)w << 26) | (sign << 31);
          if (PLANES) { rec_lo[pos] = lo; rec_hi[pos] = hi; } else recs[pos] = make_uint2(lo, hi);
          pid[pos] = (uint16_t)pr;
        }
      }
    }
    __syncthreads();
    const uint32_t total = total_sh;
    for (uint32_t i = threadIdx.x; i < total; i += T) part[delta[pid[i]] + i] = PLANES ? make_uint2(rec_lo[i], rec_hi[i]) : recs[i];
    __syncthreads();
  }
}

// one (job, partition) per trip of a grid-stride loop (capped grid: see PART_SORT_GRID): bucket offsets, final placement, and the
// size classes of the partition's buckets added to class_hist (k_border_place)
// stage_cap > 0 (round 4): a partition of at most stage_cap entries is placed in LDS first and copied out in order -- whole lines
// instead of one 4-byte store per entry into a 26-KB window (323 MB written for 54 MB of entries before); larger partitions (the
// protocol's heavy buckets, or MSMs much larger than the plan was sized for) take the direct path
__global__ __launch_bounds__(PART_SORT_THREADS) void k_part_sort(const MsmBatchDev batch, const uint2* __restrict__ part,
                                                   const uint32_t* __restrict__ base, const uint32_t* __restrict__ total, size_t hn, int P,
                                                   uint32_t jobstride, uint32_t* __restrict__ off, uint32_t* __restrict__ entries, uint32_t* __restrict__ class_hist,
                                                   uint32_t stage_cap) {
  extern __shared__ uint32_t stage[];
  __shared__ uint32_t cnt[1 << PART_LOW_BITS];
  __shared__ uint32_t sc4[256];
  __shared__ uint32_t chist[256];
  const uint32_t M = (uint32_t)batch.k * jobstride;
  const uint32_t nparts = (uint32_t)batch.k * (uint32_t)P;
  constexpr uint32_t T = PART_SORT_THREADS;
  if (threadIdx.x < 256) chist[threadIdx.x] = 0;
  for (uint32_t pi = blockIdx.x; pi < nparts; pi += gridDim.x) {
  const int job = pi / P, t = pi % P;
  const uint32_t nblk = batch.tile0[job + 1] - batch.tile0[job];
  const size_t idx = (size_t)P * batch.tile0[job] + (size_t)t * nblk, idx_next = idx + nblk;
  const uint32_t beg = idx < hn ? base[idx] : *total;
  const uint32_t end = idx_next < hn ? base[idx_next] : *total;
  const bool last_block = pi == nparts - 1;
  for (int t = threadIdx.x; t < (1 << PART_LOW_BITS); t += T) cnt[t] = 0;
  __syncthreads();
  // every lane of a wave must reach lds_take: round the trip count up to the wave
  const uint32_t span = end - beg, trips = (span + T - 1) / T;
  constexpr int B = 8;                       // loads in flight per lane (the walk is latency-bound otherwise)
  for (uint32_t k0 = 0; k0 < trips; k0 += B) {
    uint32_t lo[B];
#pragma unroll
    for (int j = 0; j < B; j++) { const uint32_t e = beg + (k0 + j) * T + threadIdx.x; lo[j] = e < end ? part[e].x : 0xffffffffu; }
#pragma unroll
    for (int j = 0; j < B; j++) if (k0 + j < trips) lds_take(cnt, lo[j] != 0xffffffffu, lo[j] & ((1u << PART_LOW_BITS) - 1));
  }
  __syncthreads();
  // exclusive scan of the counters: CPT per thread + a 256-wide scan
  constexpr int CPT = (1 << PART_LOW_BITS) / 256;
  const bool scans = threadIdx.x < 256;        // the first 256 threads scan; the others only meet them at the barriers
  uint32_t v[CPT], ssum = 0;
  for (int k = 0; k < CPT; k++) { v[k] = scans ? cnt[threadIdx.x * CPT + k] : 0u; ssum += v[k]; }
  uint32_t ex = block_exclusive_scan_256(ssum, sc4, nullptr);
  for (int k = 0; scans && k < CPT; k++) {
    const uint32_t local = (uint32_t)t * (1u << PART_LOW_BITS) + threadIdx.x * CPT + k;
    cnt[threadIdx.x * CPT + k] = ex;               // becomes the running cursor
    if (local < jobstride) { off[(uint32_t)job * jobstride + local] = beg + ex; atomicAdd(&chist[size_class(v[k])], 1u); }
    ex += v[k];
  }
  if (last_block && threadIdx.x == 0) off[M] = *total;
  __syncthreads();
  for (uint32_t k0 = 0; k0 < trips; k0 += B) {
    uint32_t lo[B], pay[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
      const uint32_t e = beg + (k0 + j) * T + threadIdx.x;
      const bool live = e < end;
      const uint2 rec = live ? part[e] : make_uint2(0xffffffffu, 0u);
      lo[j] = rec.x;
      pay[j] = rec.y;
    }
#pragma unroll
    for (int j = 0; j < B; j++) {
      if (k0 + j >= trips) continue;
      const bool live = lo[j] != 0xffffffffu;
      const uint32_t pos = lds_take(cnt, live, lo[j] & ((1u << PART_LOW_BITS) - 1));
      if (live) { if (span <= stage_cap) stage[pos] = pay[j]; else entries[beg + pos] = pay[j]; }
    }
  }
  __syncthreads();
  if (span <= stage_cap) {
    for (uint32_t i = threadIdx.x; i < span; i += T) entries[beg + i] = stage[i];
    __syncthreads();
  }
  }
  const uint32_t c = threadIdx.x < 256 ? chist[threadIdx.x] : 0u;
  if (c) atomicAdd(&class_hist[threadIdx.x], c);
}

// ---- exclusive scan of u32 (three small kernels; tile = 2048) -------------------------------
// (threads 256 and up of a wider workgroup take part in the barriers only)
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* sh, uint32_t* total) {
  const int t = threadIdx.x;
  const bool in = t < 256;
  if (in) sh[t] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    uint32_t x = in && t >= o ? sh[t - o] : 0;
    __syncthreads();
    if (in) sh[t] += x;
    __syncthreads();
  }
  uint32_t incl = in ? sh[t] : 0u;
  if (total) *total = sh[255];
  __syncthreads();
  return incl - v;
}
__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t* in, size_t m, uint32_t* tile_sums) {
  __shared__ uint32_t sh[256];
  size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x * 8;
  uint32_t s = 0;
  for (int k = 0; k < 8; k++) if (base + k < m) s += in[base + k];
  uint32_t tot;
  block_exclusive_scan_256(s, sh, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_scan_top(uint32_t* tile_sums, int ntiles, uint32_t* total_out) {
  __shared__ uint32_t sh[256];
  __shared__ uint32_t carry_sh;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 256) {
    int idx = base + threadIdx.x;
    uint32_t v = idx < ntiles ? tile_sums[idx] : 0, tot;
    uint32_t ex = block_exclusive_scan_256(v, sh, &tot);
    uint32_t carry = carry_sh;
    if (idx < ntiles) tile_sums[idx] = ex + carry;
    __syncthreads();
    if (threadIdx.x == 0) carry_sh = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_sh;
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* in, size_t m, const uint32_t* tile_sums, uint32_t* out) {
  __shared__ uint32_t sh[256];
  size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x * 8;
  uint32_t v[8], s = 0;
  for (int k = 0; k < 8; k++) { v[k] = base + k < m ? in[base + k] : 0; s += v[k]; }
  uint32_t ex = block_exclusive_scan_256(s, sh, nullptr) + tile_sums[blockIdx.x];
  for (int k = 0; k < 8; k++) { if (base + k < m) out[base + k] = ex; ex += v[k]; }
}

// ---- bucket order: largest buckets first -----------------------------------------------------
// One thread walks one bucket, so a wave runs as long as its largest bucket.  Bucket sizes are
// Poisson around N / 2^(c-1) (and 4x that in the top window): in index order a wave idles ~30 % of its
// lanes.  A counting sort of the buckets by size (256 classes) puts equal-length walks in the same wave.
// The size classes are counted by k_part_sort as it computes the bucket sizes (class_hist, cleared with the heavy-bucket header at the
// start of a chain); this kernel turns the 256 totals into class offsets (every workgroup scans them again: 256 words), reserves
// its share of every class with one atomic per non-empty class and places its buckets.  Two launches (k_border_hist + three scan
// launches + k_border_scatter before) became one; the order inside a class is arbitrary, which no result depends on.
__global__ __launch_bounds__(256) void k_border_place(const uint32_t* __restrict__ off, uint32_t nbuckets, const uint32_t* __restrict__ class_hist,
                                                      uint32_t* __restrict__ class_cursor, uint32_t* __restrict__ order) {
  __shared__ uint32_t h[256], start[256], sc[256];
  const uint32_t tot = class_hist[threadIdx.x];
  const uint32_t cls_off = block_exclusive_scan_256(tot, sc, nullptr);
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * 2048;
  uint32_t cls[8];
  for (int k = 0; k < 8; k++) {
    const uint32_t b = base + k * 256 + threadIdx.x;
    cls[k] = b < nbuckets ? size_class(off[b + 1] - off[b]) : 0xffffffffu;
    if (cls[k] != 0xffffffffu) atomicAdd(&h[cls[k]], 1u);
  }
  __syncthreads();
  const uint32_t mine = h[threadIdx.x];
  start[threadIdx.x] = cls_off + (mine ? atomicAdd(&class_cursor[threadIdx.x], mine) : 0u);
  __syncthreads();
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int k = 0; k < 8; k++)
    if (cls[k] != 0xffffffffu) order[start[cls[k]] + atomicAdd(&h[cls[k]], 1u)] = base + k * 256 + threadIdx.x;
}
// ---- bucket accumulation ---------------------------------------------------------------------
// entry = point index | sign (bit 31); over window tables additionally the window in bits 26..30 and the
// point of (i, w) is tab[w * stride + i] = 2^(c w) P_i
__device__ __forceinline__ size_t entry_point(uint32_t e, long stride) {
  if (stride == 0) return (size_t)(e & 0x7fffffffu);
  return (size_t)(e & 0x03ffffffu) + (size_t)((e >> 26) & 31u) * (size_t)stride;
}

__global__ __launch_bounds__(256, 2) void k_bucket_accum(const MsmBatchDev batch, uint32_t jobstride, const uint32_t* __restrict__ entries,
                                                      const uint32_t* __restrict__ off, const uint32_t* __restrict__ order,
                                                      uint32_t nbuckets, uint32_t heavy_t,
                                                      G1XYZZ* __restrict__ buckets, HeavyMeta* hm, HeavyRec* hrecs) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbuckets) return;
  const uint32_t b = order[t];
  const uint32_t beg = off[b], end = off[b + 1];
  const uint32_t cnt = end - beg;
  const PointArray pts{batch.points[b / jobstride], batch.pt_stride};
  const long stride = batch.tstride[b / jobstride];
  if (cnt > heavy_t) {
    const uint32_t h = atomicAdd(&hm->n_heavy, 1u);
    hrecs[h].bucket = b;
    hrecs[h].cnt = cnt;
    return;
  }
  // Two-deep software pipeline: the index of entry e+2 and the point of entry e+1 are in flight while
  // entry e is added (the gather is a dependent load pair, ~2 us of HBM latency per entry otherwise;
  // only two waves per SIMD fit, so the hardware cannot hide it by itself).
  G1XYZZ acc = G1XYZZ::inf();
  if (cnt) {
    const uint32_t last = end - 1;
    // The first entry only initialises the accumulator (the fused addition would run all its products with every lane
    // switched off: one wasted addition per bucket, 4 % of the kernel at 26 entries per bucket, 7 % at 15) ...
    // ... and the second entry meets an affine accumulator (ZZ = ZZZ = 1): four of the ten products are trivial
    // (g1_add_affine_walk).
    const uint32_t e_first = entries[beg];
    const uint32_t e_second = entries[beg + 1 <= last ? beg + 1 : last];
    uint32_t e_cur = entries[beg + 2 <= last ? beg + 2 : last];
    uint32_t e_nxt = entries[beg + 3 <= last ? beg + 3 : last];
    G1Affine p_first = pts[entry_point(e_first, stride)];
    G1Affine p_second = pts[entry_point(e_second, stride)];
    G1Affine p_cur = pts[entry_point(e_cur, stride)];
    if (e_first >> 31) p_first.y = fp_neg(p_first.y);
    if (e_second >> 31) p_second.y = fp_neg(p_second.y);
    acc = g1_add_affine_walk(p_first, p_second, cnt < 2);
    for (uint32_t e = beg + 2; e < end; e++) {
      const uint32_t e_nn = entries[e + 2 <= last ? e + 2 : last];
      const G1Affine p_nxt = pts[entry_point(e_nxt, stride)];
      acc = g1_add_mixed_walk(acc, p_cur, e_cur >> 31);          // (bit 31 of an entry: the digit's sign)
      p_cur = p_nxt; e_cur = e_nxt; e_nxt = e_nn;
    }
  }
  buckets[b] = acc;
}

// The same walk with T = 2 or 4 lanes per bucket (round 6): lane r takes entries r, r + T, r + 2T, .. of its bucket and the
// T partial sums are folded with whole-point exchanges over DPP (one or two full additions, every lane of the group computes them).  For
// launches that would leave the chip under-filled with one lane per bucket: a stand-alone MSM over a small SRS is ONE job of 2^15 or 2^16
// buckets -- 512 or 1024 waves for 2048 wave slots, each walking 30-60 entries at the latency of a lone wave (18.5 us per addition) --
// while four lanes per bucket are 2048-4096 waves with walks a quarter as long.  (Inside prove() such plans run as one chain of 15 jobs
// and fill the chip by themselves.)  The entries are walked with a one-deep prefetch; exceptional additions fall back as in the kernel above.
template <int CTRL>
__device__ __forceinline__ G1XYZZ g1_point_quad_perm(const G1XYZZ& p) {
  G1XYZZ r;
  r.x = fq_quad<CTRL>(p.x); r.y = fq_quad<CTRL>(p.y); r.zz = fq_quad<CTRL>(p.zz); r.zzz = fq_quad<CTRL>(p.zzz);
  return r;
}
template <int T>
__global__ __launch_bounds__(256, 2) void k_bucket_accum_split(const MsmBatchDev batch, uint32_t jobstride, const uint32_t* __restrict__ entries,
                                                            const uint32_t* __restrict__ off, const uint32_t* __restrict__ order,
                                                            uint32_t nbuckets, uint32_t heavy_t,
                                                            G1XYZZ* __restrict__ buckets, HeavyMeta* hm, HeavyRec* hrecs) {
  static_assert(T == 2 || T == 4, "two or four lanes per bucket (a quad holds whole groups)");
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = t / T, r = t % T;
  if (g >= nbuckets) return;                          // (whole groups leave together: T divides the wave)
  const uint32_t b = order[g];
  const uint32_t beg = off[b], end = off[b + 1];
  const uint32_t cnt = end - beg;
  if (cnt > heavy_t) {                                // recorded once; k_heavy_accum walks it
    if (r == 0) { const uint32_t h = atomicAdd(&hm->n_heavy, 1u); hrecs[h].bucket = b; hrecs[h].cnt = cnt; }
    return;
  }
  const PointArray pts{batch.points[b / jobstride], batch.pt_stride};
  const long stride = batch.tstride[b / jobstride];
  G1XYZZ acc = G1XYZZ::inf();
  uint32_t e = beg + r;
  if (e < end) {
    uint32_t e_cur = entries[e];
    G1Affine p_cur = pts[entry_point(e_cur, stride)];
    if (e_cur >> 31) p_cur.y = fp_neg(p_cur.y);
    acc = G1XYZZ::from_affine(p_cur);                 // the lane's first entry is a copy, not an addition
    e += T;
    if (e < end) {
      e_cur = entries[e];
      p_cur = pts[entry_point(e_cur, stride)];
      for (; e < end; e += T) {
        const uint32_t e_n = e + T < end ? e + T : e;
        const uint32_t e_nxt = entries[e_n];
        const G1Affine p_nxt = pts[entry_point(e_nxt, stride)];
        acc = g1_add_mixed_walk(acc, p_cur, e_cur >> 31);
        p_cur = p_nxt; e_cur = e_nxt;
      }
    }
  }
  constexpr int QP_SWAP1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);     // [1,0,3,2]
  if (T == 4) acc = g1_add(acc, g1_point_quad_perm<QP_SWAP2>(acc));   // lanes 0, 2: s0 + s2; lanes 1, 3: s1 + s3
  acc = g1_add(acc, g1_point_quad_perm<QP_SWAP1>(acc));               // every lane of the group: the bucket's sum
  if (r == 0) buckets[b] = acc;
}

// exclusive scan over the HEAVY_THREADS lanes of a workgroup (wave shuffles + one word per wave in LDS); *total = the sum
__device__ __forceinline__ uint32_t heavy_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t before = 0, tot = 0;
  for (int i = 0; i < HEAVY_THREADS / 64; i++) { const uint32_t w = wsum[i]; if (i < wv) before += w; tot += w; }
  __syncthreads();
  *total = tot;
  return before + incl - v;
}

struct HeavyWork { uint32_t bucket, skip, len, slot; };     // entries [off[bucket] + skip, + len) -> partial[slot]

__global__ __launch_bounds__(HEAVY_THREADS, 1) void k_heavy_accum(const MsmBatchDev batch, uint32_t jobstride, const uint32_t* __restrict__ entries,
                                                     const uint32_t* __restrict__ off, const HeavyMeta* hm,
                                                     HeavyRec* hrecs, G1XYZZ* __restrict__ partial) {
  constexpr int HALF = HEAVY_THREADS / 2;
  __shared__ G1XYZZ sh[HALF];                // 48 KB: the upper half of the workgroup hands its sums to the lower half first
  __shared__ HeavyWork work[HEAVY_THREADS];
  __shared__ uint32_t wsum[HEAVY_THREADS / 64];
  __shared__ uint32_t n_work;
  const uint32_t nh = hm->n_heavy;
  if (nh == 0) return;
  // how many entries there are, hence how long a stretch is
  uint32_t mine = 0, total;
  for (uint32_t r = threadIdx.x; r < nh; r += HEAVY_THREADS) mine += hrecs[r].cnt;
  heavy_scan(mine, wsum, &total);
  uint32_t stretch = (total + gridDim.x - 1) / gridDim.x;
  if (stretch < HEAVY_MIN_STRETCH) stretch = HEAVY_MIN_STRETCH;
  const uint64_t lo = (uint64_t)blockIdx.x * stretch, hi = lo + stretch;     // this workgroup's stretch of the concatenated entries
  if (lo >= total) return;
  if (threadIdx.x == 0) n_work = 0;
  uint32_t carry_cnt = 0, carry_np = 0;
  for (uint32_t r0 = 0; r0 < nh && carry_cnt < hi; r0 += HEAVY_THREADS) {
    __syncthreads();
    const uint32_t r = r0 + threadIdx.x;
    const HeavyRec rec = r < nh ? hrecs[r] : HeavyRec{0u, 0u, 0u, 0u};
    uint32_t batch_cnt, batch_np;
    const uint64_t start = (uint64_t)carry_cnt + heavy_scan(rec.cnt, wsum, &batch_cnt), end = start + rec.cnt;
    const uint32_t first_g = (uint32_t)(start / stretch), last_g = rec.cnt ? (uint32_t)((end - 1) / stretch) : first_g;
    const uint32_t np = rec.cnt ? last_g - first_g + 1 : 0u;
    const uint32_t base = carry_np + heavy_scan(np, wsum, &batch_np);
    if (rec.cnt && blockIdx.x >= first_g && blockIdx.x <= last_g) {
      const uint64_t a = start > lo ? start : lo, b = end < hi ? end : hi;
      work[atomicAdd(&n_work, 1u)] = HeavyWork{rec.bucket, (uint32_t)(a - start), (uint32_t)(b - a), base + (blockIdx.x - first_g)};
      if (blockIdx.x == first_g) { hrecs[r].base = base; hrecs[r].npart = np; }
    }
    carry_cnt += batch_cnt;
    carry_np += batch_np;
    __syncthreads();
    const uint32_t nw = n_work;
    for (uint32_t wi = 0; wi < nw; wi++) {
      const HeavyWork wk = work[wi];
      const PointArray pts{batch.points[wk.bucket / jobstride], batch.pt_stride};
      const long stride = batch.tstride[wk.bucket / jobstride];
      const uint32_t beg = off[wk.bucket] + wk.skip, end_e = beg + wk.len;
      G1XYZZ acc = G1XYZZ::inf();
      if (beg + threadIdx.x < end_e) {
        const uint32_t last = beg + threadIdx.x + ((end_e - 1 - beg - threadIdx.x) / HEAVY_THREADS) * HEAVY_THREADS;   // this lane's last entry
        uint32_t e = beg + threadIdx.x;
        const uint32_t e_first = entries[e];
        G1Affine p_first = pts[entry_point(e_first, stride)];
        if (e_first >> 31) p_first.y = fp_neg(p_first.y);
        acc = G1XYZZ::from_affine(p_first);                       // as in k_bucket_accum: the first entry is a copy, not an addition
        e += HEAVY_THREADS;
        uint32_t e_cur = entries[e <= last ? e : last];
        uint32_t e_nxt = entries[e + HEAVY_THREADS <= last ? e + HEAVY_THREADS : last];
        G1Affine p_cur = pts[entry_point(e_cur, stride)];
        for (; e < end_e; e += HEAVY_THREADS) {
          const uint32_t e_nn = entries[e + 2 * HEAVY_THREADS <= last ? e + 2 * HEAVY_THREADS : last];
          const G1Affine p_nxt = pts[entry_point(e_nxt, stride)];
          acc = g1_add_mixed_walk(acc, p_cur, e_cur >> 31);
          p_cur = p_nxt; e_cur = e_nxt; e_nxt = e_nn;
        }
      }
      // 512 sums -> 1: one whole addition per lane of the lower half, then quads
      if ((int)threadIdx.x >= HALF) sh[threadIdx.x - HALF] = acc;
      __syncthreads();
      if ((int)threadIdx.x < HALF) acc = g1_add(acc, sh[threadIdx.x]);
      __syncthreads();
      if ((int)threadIdx.x < HALF) sh[threadIdx.x] = acc;
      __syncthreads();
      const int cr = threadIdx.x & 3, q = threadIdx.x >> 2;
      for (int sp = HALF / 2; sp >= 1; sp >>= 1) {
        if (q < sp) {
          Fq* pa = reinterpret_cast<Fq*>(&sh[q]) + cr;
          const Fq* pb = reinterpret_cast<const Fq*>(&sh[q + sp]) + cr;
          *pa = g1q_add(*pa, *pb, cr);
        }
        __syncthreads();
      }
      if (threadIdx.x < 4) reinterpret_cast<Fq*>(&partial[wk.slot])[threadIdx.x] = reinterpret_cast<const Fq*>(&sh[0])[threadIdx.x];
      __syncthreads();
    }
    if (threadIdx.x == 0) n_work = 0;
  }
}

// one 64-lane workgroup per heavy bucket: its 16 quads stride over the bucket's partials, then a tree of quads through LDS
__global__ __launch_bounds__(64, 2) void k_heavy_finish(const HeavyMeta* hm, const HeavyRec* hrecs, const G1XYZZ* __restrict__ partial,
                                                     G1XYZZ* __restrict__ buckets) {
  __shared__ G1XYZZ sh[16];
  const uint32_t n_heavy = hm->n_heavy;
  const int cr = threadIdx.x & 3, q = threadIdx.x >> 2;
  for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
    const HeavyRec r = hrecs[h];
    Fq acc = Fq::zero();                                       // coordinate cr of the point at infinity
    for (uint32_t k = q; k < r.npart; k += 16) {
      const Fq c = reinterpret_cast<const Fq*>(&partial[r.base + k])[cr];
      acc = k == (uint32_t)q ? c : g1q_add(acc, c, cr);
    }
    reinterpret_cast<Fq*>(&sh[q])[cr] = acc;
    __syncthreads();
    for (int sp = 8; sp >= 1; sp >>= 1) {
      if (q < sp && (uint32_t)(q + sp) < r.npart) {
        Fq* pa = reinterpret_cast<Fq*>(&sh[q]) + cr;
        *pa = g1q_add(*pa, reinterpret_cast<const Fq*>(&sh[q + sp])[cr], cr);
      }
      __syncthreads();
    }
    if (threadIdx.x < 4) reinterpret_cast<Fq*>(&buckets[r.bucket])[threadIdx.x] = reinterpret_cast<const Fq*>(&sh[0])[threadIdx.x];
    __syncthreads();
  }
}

// ---- bucket reduction: sum_b (b+1) * B_b per window -----------------------------------------
// sg_base: the set's first segment is segment sg_base of a longer bucket sequence (one rank's bucket range of an MSM whose buckets are
// sharded across ranks, msm_reduce_slices_enqueue): bucket j of segment sg weighs (sg_base + sg) K + j + 1
//
// Registers: two XYZZ operands, the result and the temporaries of a full addition around the product routine's own 75 registers
// are ~320 VGPRs when both running sums live in registers: one wave per SIMD.  `tot` is touched once per bucket, so it lives in
// LDS (48 words per thread, word-major: conflict-free) while `run` takes the bucket in; the kernel then fits two waves per SIMD.
#ifndef SONIC_SEGMENTS_LDS_TOT
#define SONIC_SEGMENTS_LDS_TOT 1
#endif
struct LdsPoint {
  uint32_t* base;      // &sh[threadIdx.x]; word w at base[w * 256]
  __device__ __forceinline__ void store(const G1XYZZ& p) const {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&p);
#pragma unroll
    for (int k = 0; k < 48; k++) base[k * 256] = w[k];
  }
  __device__ __forceinline__ G1XYZZ load() const {
    G1XYZZ p;
    uint32_t* w = reinterpret_cast<uint32_t*>(&p);
#pragma unroll
    for (int k = 0; k < 48; k++) w[k] = base[k * 256];
    return p;
  }
};
__global__ __launch_bounds__(256, SONIC_SEGMENTS_LDS_TOT ? 2 : 1) void k_bucket_segments(const G1XYZZ* __restrict__ buckets, int W, int NB, int K, int nseg, int sg_base,
                                                        G1XYZZ* __restrict__ segres) {
#if SONIC_SEGMENTS_LDS_TOT
  __shared__ uint32_t tot_sh[48 * 256];
  const LdsPoint tl{tot_sh + threadIdx.x};
#endif
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= W * nseg) return;
  const int w = t / nseg, sl = t % nseg;
  const G1XYZZ* B = buckets + (size_t)w * NB + (size_t)sl * K;
  const int sg = sg_base + sl;
  const int k0 = sg * K;
  G1XYZZ run = G1XYZZ::inf();
#if SONIC_SEGMENTS_LDS_TOT
  tl.store(G1XYZZ::inf());
  for (int j = K - 1; j >= 0; j--) {
    run = g1_add(run, B[j]);
    tl.store(g1_add(tl.load(), run));          // ends as sum_j (j+1) B[j]
  }
#else
  G1XYZZ tot = G1XYZZ::inf();
  for (int j = K - 1; j >= 0; j--) {
    run = g1_add(run, B[j]);
    tot = g1_add(tot, run);          // ends as sum_j (j+1) B[j]
  }
#endif
  G1XYZZ acc = G1XYZZ::inf();
  bool have_multiple = false;
  if (nseg % 64 == 0 && sg_base % 64 == 0) {
    // k0 * run with k0 = sg * K: the 64 lanes of a wave hold consecutive sg, so the bits of sg above the lane bits are
    // wave-uniform and their conditional additions are uniform branches; only the six lane bits pay for both paths
    // (a per-lane double-and-add executes an addition at every bit as soon as any lane needs one).
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)sg >> 6), lo = (uint32_t)sg & 63u;
    for (int i = 31 - __clz((int)(hi | 1u)); i >= 0; i--) {
      acc = g1_dbl(acc);
      if ((hi >> i) & 1u) acc = g1_add(acc, run);
    }
    for (int i = 5; i >= 0; i--) {
      acc = g1_dbl(acc);
      if ((lo >> i) & 1u) acc = g1_add(acc, run);
    }
    for (int s = K; s > 1; s >>= 1) acc = g1_dbl(acc);
    have_multiple = true;
  } else if (k0) {
    acc = g1_mul_small(run, (uint32_t)k0);
    have_multiple = true;
  }
#if SONIC_SEGMENTS_LDS_TOT
  G1XYZZ tot = tl.load();
#endif
  if (have_multiple) tot = g1_add(tot, acc);
  segres[t] = tot;
}

// block = bucket set (job * Wb + window); the sum lands in that job's slot
__global__ __launch_bounds__(256, 2) void k_window_sum(const G1XYZZ* __restrict__ segres, int W, int c, int nseg, const MsmBatchDev batch) {
  __shared__ G1XYZZ sh[256];
  const int w = blockIdx.x % W;
  MsmSlot* slot = batch.slot[blockIdx.x / W];
  segres += (size_t)(blockIdx.x - w) * nseg;
  G1XYZZ acc = G1XYZZ::inf();
  for (int s = threadIdx.x; s < nseg; s += 256) acc = g1_add(acc, segres[(size_t)w * nseg + s]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = g1_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    G1XYZZ r = sh[0];                 // leaves the device for the host tail: representatives below q (field.hpp, lazy range)
    r.x = fp_canonical(r.x); r.y = fp_canonical(r.y); r.zz = fp_canonical(r.zz); r.zzz = fp_canonical(r.zzz);
    slot->win[w] = r;
    if (w == 0) { slot->W = W; slot->c = c; slot->pad0 = 0; slot->pad1 = 0; }
  }
}

// out[g] = sum of in[g * group .. (g+1) * group): pre-reduction when one window has very many segments
__global__ __launch_bounds__(256, 2) void k_group_sum(const G1XYZZ* __restrict__ in, long n_in, int group, G1XYZZ* __restrict__ out) {
  __shared__ G1XYZZ sh[256];
  const long base = (long)blockIdx.x * group;
  G1XYZZ acc = G1XYZZ::inf();
  for (long s = threadIdx.x; s < group && base + s < n_in; s += 256) acc = g1_add(acc, in[base + s]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] = g1_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// ---- bucket reduction as an in-place bit-sum butterfly (round 4; shared bucket sets) ---------------------------------------------
// sum_i (i + 1) B_i over a set of 2^L buckets  =  tot + sum_j 2^j bs_j   with  tot = sum_i B_i,  bs_j = sum_{i : bit j of i} B_i.
// The L + 1 sums come out of ONE log-depth tree with no scalar multiplication in it: for a block of 2^m buckets starting at s keep
// Z[s] = the block's total and Z[s + 2^j] = the block's sum over local indices with bit j set (j < m).  Two neighbouring blocks
// (s and s + h, h = 2^(m-1)) combine IN PLACE with m additions that touch disjoint elements:
//     Z[s] += Z[s + h]              Z[s + 2^j] += Z[s + h + 2^j]  (j < m - 1)
// and the new bs_(m-1) is the right block's total, which already sits at Z[s + h] = Z[s + 2^(m-1)].  Level m costs (NB / 2^m) m
// additions, all independent: ~2 NB additions in total -- fewer than the running sums over K-bucket segments, which paid 2 NB + a
// ~30-step multiple per segment (2.5 NB at K = 64, 5.8 NB at K = 8) -- and the dependent chain is L additions deep instead of
// 2K + ~30 + 16.  What is left for the host is Horner over L + 1 points (msm_finish_host: ~20 us), next to the normalisation it
// does anyway.  Measured: profiles/r04_bucket_tree.txt.
constexpr int TREE_BLOCK_LOG = 10;          // buckets per workgroup of the first launch: 1024 (4 per thread)
constexpr int TREE_MID_LOG = 14;            // second launch: one workgroup per 2^14 buckets, levels 11..14; third: one per set
#ifndef SONIC_TREE_LATENCY_MAX
#define SONIC_TREE_LATENCY_MAX (1L << 17)
#endif
constexpr long TREE_LATENCY_MAX_BUCKETS = SONIC_TREE_LATENCY_MAX;     // up to here (all sets of a launch together) the all-quad form below
constexpr int TREE_LATENCY_BLOCK_LOG = 7, TREE_LATENCY_MID_LOG = 12;

// item `it` of level m inside the span that starts at s0: pair = it / m, component = it % m (0: total, j: bit j - 1)
__device__ __forceinline__ void tree_item(G1XYZZ* __restrict__ Z, size_t s0, int m, uint32_t it) {
  const uint32_t pair = it / (uint32_t)m, comp = it % (uint32_t)m;
  const size_t s = s0 + ((size_t)pair << m);
  const size_t pos = comp == 0 ? 0 : (size_t)1 << (comp - 1);
  const size_t h = (size_t)1 << (m - 1);
  Z[s + pos] = g1_add(Z[s + pos], Z[s + h + pos]);
}

// the same item shared by the four lanes of a quad (g1_quad.hpp): lane r moves coordinate r.  Every lane of the quad calls it with
// the same `it`.
__device__ __forceinline__ void tree_item_quad(G1XYZZ* __restrict__ Z, size_t s0, int m, uint32_t it, int r) {
  const uint32_t pair = it / (uint32_t)m, comp = it % (uint32_t)m;
  const size_t s = s0 + ((size_t)pair << m);
  const size_t pos = comp == 0 ? 0 : (size_t)1 << (comp - 1);
  const size_t h = (size_t)1 << (m - 1);
  Fq* pa = reinterpret_cast<Fq*>(&Z[s + pos]) + r;
  const Fq* pb = reinterpret_cast<const Fq*>(&Z[s + h + pos]) + r;
  *pa = g1q_add(*pa, *pb, r);
}
// one level inside a workgroup: whole additions per lane while there are more additions than lanes, a quad per addition once at most
// two rounds of quads cover the level (the chain, not the issue slots, is what the late levels wait for)
__device__ __forceinline__ void tree_level(G1XYZZ* __restrict__ Z, size_t s0, int m, uint32_t items, int quads) {
  // quads = how many rounds of quads a level may take before whole additions per lane are preferred (0: never)
  if (quads > 0 && items * 4 <= (uint32_t)quads * blockDim.x) {
    const int r = threadIdx.x & 3;
    for (uint32_t it = threadIdx.x >> 2; it < items; it += blockDim.x >> 2) tree_item_quad(Z, s0, m, it, r);
  } else {
    for (uint32_t it = threadIdx.x; it < items; it += blockDim.x) tree_item(Z, s0, m, it);
  }
}

// first launch: levels 1, 2 in registers (4 buckets per thread: 4 additions), then levels 3 .. LB through memory, one workgroup per
// block of 2^LB buckets (the block's lines stay in this CU's cache; __syncthreads orders the levels)
__global__ __launch_bounds__(256, 2) void k_bucket_tree_block(G1XYZZ* __restrict__ Z, int LB, int quads) {
  const size_t s0 = (size_t)blockIdx.x << LB;
  const uint32_t nthreads = 1u << (LB - 2);
  const bool leaf_in_registers = quads < 32;
  if (!leaf_in_registers) {
    tree_level(Z, s0, 1, 1u << (LB - 1), quads);
    __syncthreads();
    tree_level(Z, s0, 2, 2u << (LB - 2), quads);
  } else if (threadIdx.x < nthreads) {
    // (ordered so that at most two points are live beside an addition's own temporaries: with b1 and b3 held across the first three
    // additions the kernel spilled 304 registers under its two-waves-per-SIMD bound)
    G1XYZZ* b = Z + s0 + 4 * (size_t)threadIdx.x;
    const G1XYZZ odd = g1_add(b[1], b[3]);        // bit 0
    const G1XYZZ p23 = g1_add(b[2], b[3]);        // bit 1
    b[3] = odd;                                   // (b[3] is dead after this level: parked here until b[1] may be overwritten)
    const G1XYZZ tot = g1_add(g1_add(b[0], b[1]), p23);
    b[2] = p23;
    b[0] = tot;
    b[1] = b[3];
  }
  for (int m = 3; m <= LB; m++) {
    __syncthreads();
    tree_level(Z, s0, m, (uint32_t)m << (LB - m), quads);
  }
}

// later launches: levels m_lo .. m_hi over spans of 2^span_log buckets, one workgroup per span.  final_L > 0 (then span_log == final_L:
// one workgroup per bucket set): the set's sums go to its job's slot -- win[j] = bs_j (j < L), win[L] = total, W = L, c = 1,
// pad0 = the total's weight (1, or base + 1 for a bucket range that starts at `base`), pad1 = 1 marks the form (msm_finish_host)
template <int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 256 ? 2 : 1) void k_bucket_tree_levels(G1XYZZ* __restrict__ Z, int span_log, int m_lo, int m_hi, int final_L,
                                                               uint32_t tot_mul, int quads, const MsmBatchDev batch) {
  const size_t s0 = (size_t)blockIdx.x << span_log;
  for (int m = m_lo; m <= m_hi; m++) {
    if (m > m_lo) __syncthreads();
    tree_level(Z, s0, m, (uint32_t)m << (span_log - m), quads);
  }
  if (final_L > 0) {
    __syncthreads();
    MsmSlot* slot = batch.slot[blockIdx.x];
    const int off = batch.slot_off[blockIdx.x];
    if ((int)threadIdx.x <= final_L) {
      G1XYZZ r = Z[s0 + ((int)threadIdx.x == final_L ? 0 : (size_t)1 << threadIdx.x)];
      r.x = fp_canonical(r.x); r.y = fp_canonical(r.y); r.zz = fp_canonical(r.zz); r.zzz = fp_canonical(r.zzz);   // leaves the device (lazy range)
      slot->win[off + threadIdx.x] = r;
    }
    if (threadIdx.x == 0 && off == 0) { slot->W = final_L; slot->c = 1; slot->pad0 = (int)tot_mul; slot->pad1 = batch.slot_form; }
  }
}


// reduces `sets` bucket sets of 2^L buckets each (consecutive in Z) into the jobs' slots
static void bucket_tree_enqueue(hipStream_t st, G1XYZZ* Z, int sets, int L, uint32_t tot_mul, const MsmBatchDev& batch) {
  if (((long)sets << L) <= TREE_LATENCY_MAX_BUCKETS) {
    // Few buckets (round 6): the chip has more lanes than the tree has additions on ANY level -- level 1 of 2^16 buckets is 32768
    // additions against 131072 lanes at two waves per SIMD -- so every level runs on quads (a quad's addition is 5 products deep instead
    // of 14: g1_quad.hpp) and the blocks are small enough to spread the early levels over the whole chip: 128 buckets per workgroup
    // (one round of 64 quads per level), then 512-lane workgroups over spans of 2^12, then one per set.  The form above -- 1024-bucket
    // blocks whose four leaf additions a lane runs one after the other, whole additions while a level has more of them than the
    // workgroup has lanes -- is the one that does least WORK, which is what counts when the bucket sets of a batch fill the chip.
    // One 2^16-bucket set (an eighth of a bucket-sharded MSM's buckets, a stand-alone small MSM): 0.29 -> see profiles/r06_tree_latency.txt
    constexpr int all_quads = 1 << 20;
    const int LB = L < TREE_LATENCY_BLOCK_LOG ? L : TREE_LATENCY_BLOCK_LOG;
    LAUNCH(k_bucket_tree_block, (uint32_t)sets << (L - LB), 256, 0, st, Z, LB, all_quads);
    int done = LB;
    if (L > TREE_LATENCY_MID_LOG) {
      LAUNCH(k_bucket_tree_levels<512>, (uint32_t)sets << (L - TREE_LATENCY_MID_LOG), 512, 0, st, Z, TREE_LATENCY_MID_LOG, done + 1, TREE_LATENCY_MID_LOG, 0, 0u, all_quads, batch);
      done = TREE_LATENCY_MID_LOG;
    }
    LAUNCH(k_bucket_tree_levels<512>, (uint32_t)sets, 512, 0, st, Z, L, done + 1, L, L, tot_mul, all_quads, batch);
    return;
  }
  const int LB = L < TREE_BLOCK_LOG ? L : TREE_BLOCK_LOG;
  // (a first launch with 8 buckets per thread -- 11 additions in registers, blocks of 2048, one wave per SIMD -- measured the same:
  // 0.51 + 2 x 0.09 ms against 0.52 + 2 x 0.10 ms at 2^19 buckets)
  // rounds of quads a level may take before whole additions per lane are preferred (measured in round 4, first launch at 2^19 buckets:
  // 0 / 2 / 4 / 8 / 16 rounds 0.521 / 0.426 / 0.453 / 0.463 / 0.465 ms: the early levels are bound by issue, where a quad's 20 lane-products
  // per addition lose to a lane's 14)
  constexpr int quads = 2;
  // (round 6: 512 lanes per 1024-bucket block with TWO buckets per lane at the leaf -- twice the waves, every level one round -- measured
  // slower, 0.476 against 0.437 ms at 2^19 buckets: the levels after the second move ~300 MB of 192-byte points at power-of-two strides
  // through L2 and that traffic, not the additions' latency, is what the block kernel waits for; profiles/r06_ab_tree.txt)
  LAUNCH(k_bucket_tree_block, (uint32_t)sets << (L - LB), 256, 0, st, Z, LB, quads);
  int done = LB;
  if (L > done) {
    const int mid = L < TREE_MID_LOG ? L : TREE_MID_LOG;
    if (mid < L) {
      LAUNCH(k_bucket_tree_levels<256>, (uint32_t)sets << (L - mid), 256, 0, st, Z, mid, done + 1, mid, 0, 0u, quads, batch);
      done = mid;
    }
  }
  // last launch: one workgroup per set finishes the levels that are left (none when L <= TREE_BLOCK_LOG) and fills the slot
  LAUNCH(k_bucket_tree_levels<256>, (uint32_t)sets, 256, 0, st, Z, L, done + 1, L, L, tot_mul, quads, batch);
}

// ---- tail (host) -----------------------------------------------------------------------------
// What is left of an MSM after the bulk kernels is W <= 64 window sums: Horner over the windows
// (255 dependent doublings) and one Fq inversion for the canonical affine form.  That is ~2600
// strictly sequential Fq products -- ~10 ms on one GPU lane (measured), ~0.5 ms on a host core --
// so the host layer finishes it, with the same limb code (g1.hpp compiles for both sides).
G1XYZZ msm_finish_host(const MsmSlot& s) {
  // slots usually sit in pinned host memory, which the CPU reads uncached: limb-by-limb arithmetic straight out of it cost
  // ~20 us per window sum (0.3 ms per proof), so every window sum is copied out once before it is used
  const int W = s.W, c = s.c;
  G1XYZZ acc = G1XYZZ::inf();
  if (s.pad1 == 1 || s.pad1 == 2) {
    // bit-sum form (k_bucket_tree_levels): sum_j 2^j win[off + j] + pad0 * win[off + W]
    auto bitsum = [&](int off) {
      G1XYZZ a = G1XYZZ::inf();
      for (int w = W - 1; w >= 0; w--) {
        G1XYZZ win;
        memcpy(&win, &s.win[off + w], sizeof win);
        a = g1_add(g1_dbl(a), win);
      }
      G1XYZZ tot;
      memcpy(&tot, &s.win[off + W], sizeof tot);
      const uint32_t mul = (uint32_t)s.pad0;
      return g1_add(a, mul == 1 ? tot : g1_mul_small(tot, mul));
    };
    if (s.pad1 == 1) return bitsum(0);
    return g1_add(bitsum(0), g1_endo(bitsum(MSM_ENDO_SLOT_OFFSET)));         // endomorphism pair: sum(s1 P) + phi(sum(s2 P))
  }
  for (int w = W - 1; w >= 0; w--) {
    G1XYZZ win;
    memcpy(&win, &s.win[w], sizeof win);
    if (w == W - 1) { acc = win; continue; }
    for (int j = 0; j < c; j++) acc = g1_dbl(acc);
    acc = g1_add(acc, win);
  }
  return acc;
}
// ---------------------------------------------------------------------------------------------
bool msm_can_batch(const MsmPlan& pl) { return pl.Wb == 1 && pl.NB >= (1 << PART_LOW_BITS); }

// s -> (s mod lambda, s div lambda) for every scalar of a job (endo.hpp); standard form out, upper halves zero
__global__ __launch_bounds__(256) void k_endo_split(const Fr* __restrict__ sc, long n, int mont, Fr* __restrict__ s1, Fr* __restrict__ s2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr s = sc[i];
  if (mont) s = fp_from_mont(s);
  Fr a, b;
  endo_split(s, a, b);
  s1[i] = a; s2[i] = b;
}

void msm_enqueue_batch(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, const MsmJob* jobs, int k, bool scalars_mont, G1XYZZ* ext_buckets) {
  if (k < 1 || k > MSM_MAX_JOBS) throw std::runtime_error("msm_enqueue_batch: 1..MSM_MAX_JOBS jobs");
  if (ext_buckets && (k != 1 || pl.Wb != 1)) throw std::runtime_error("msm_enqueue_batch: bucket hand-off needs one job over a shared bucket set");
  if (k > 1 && !msm_can_batch(pl)) throw std::runtime_error("msm_enqueue_batch: plan cannot be batched");
  MsmJob split_jobs[MSM_MAX_JOBS];
  bool fold = pl.fold;
  if (pl.endo) {
    // every job becomes two device jobs over the same points: the halves s1, s2 of its scalars (endo.hpp); both sums land in the
    // job's slot (the second at win[MSM_ENDO_SLOT_OFFSET ..]) and the host adds phi(second) to the first
    if (ext_buckets) { set_error("the bucket hand-off of a bucket-sharded MSM needs the full window tables (this SRS holds endomorphism tables)"); throw HipFail{SONIC_ERR_INVALID_ARG}; }
    if (2 * k > MSM_MAX_JOBS) {
      const int h = MSM_MAX_JOBS / 2;
      for (int j0 = 0; j0 < k; j0 += h) msm_enqueue_batch(st, ws, pl, jobs + j0, k - j0 < h ? k - j0 : h, scalars_mont, nullptr);
      return;
    }
    long n_all = 0;
    for (int j = 0; j < k; j++) n_all += jobs[j].n;
    ws.endo_scalars.ensure(sizeof(Fr) * 2 * (size_t)(n_all > 0 ? n_all : 1));
    Fr* halves = ws.endo_scalars.as<Fr>();
    for (int j = 0; j < k; j++) {
      Fr* s1 = halves; Fr* s2 = halves + jobs[j].n;
      halves += 2 * jobs[j].n;
      if (jobs[j].n > 0) LAUNCH(k_endo_split, ceil_div(jobs[j].n, 256), 256, 0, st, jobs[j].scalars, jobs[j].n, (int)scalars_mont, s1, s2);
      split_jobs[2 * j] = MsmJob{jobs[j].points, s1, jobs[j].n, jobs[j].slot, jobs[j].table_stride};
      split_jobs[2 * j + 1] = MsmJob{jobs[j].points, s2, jobs[j].n, jobs[j].slot, jobs[j].table_stride};
    }
    jobs = split_jobs;
    k = 2 * k;
    scalars_mont = false;
    fold = false;             // the halves are non-negative and below 2^128: nothing to fold, no carry out of the 130 bits
  }
  for (int j = 0; j < k; j++) {
    const bool tables = pl.table_stride != 0;
    if (jobs[j].n >= (tables ? MSM_TABLE_MAX_TERMS : MSM_MAX_TERMS) || (tables && pl.W > MSM_TABLE_MAX_WINDOWS) || pl.W > MSM_MAX_WINDOWS) {
      set_error("MSM of %ld terms over %d windows does not fit the entry encoding (%s)", jobs[j].n, pl.W,
                tables ? "window tables: < 2^26 terms, <= 32 windows" : "< 2^31 terms, <= 64 windows");
      throw HipFail{SONIC_ERR_INVALID_ARG};
    }
  }
  MsmBatchDev batch;
  memset(&batch, 0, sizeof batch);
  batch.k = k;
  batch.bits = pl.bits;
  batch.slot_form = pl.endo ? 2 : 1;
  for (int j = 0; j < k; j++) batch.slot_off[j] = (pl.endo && (j & 1)) ? MSM_ENDO_SLOT_OFFSET : 0;
  batch.pt_stride = jobs[0].points.stride;
  long n_total = 0;
  for (int j = 0; j < k; j++) {
    if (jobs[j].points.stride != jobs[0].points.stride) throw std::runtime_error("msm_enqueue_batch: the jobs of a batch must share the point stride");
    batch.points[j] = jobs[j].points.p; batch.scalars[j] = jobs[j].scalars; batch.n[j] = jobs[j].n; batch.slot[j] = jobs[j].slot;
    batch.tstride[j] = pl.table_stride == 0 ? 0 : (jobs[j].table_stride ? jobs[j].table_stride : pl.table_stride);
    batch.tile0[j + 1] = batch.tile0[j] + (uint32_t)ceil_div(jobs[j].n, PART_TILE);
    n_total += jobs[j].n;
  }
  if (batch.tile0[k] == 0) for (int j = 1; j <= k; j++) batch.tile0[j] = 1;      // nothing to do: one idle workgroup keeps the chain uniform
  ws.reserve(n_total > 0 ? n_total : 1, pl, k);
  G1XYZZ* const buckets = ext_buckets ? ext_buckets : ws.buckets.as<G1XYZZ>();
  const uint32_t jobstride = (uint32_t)pl.Wb * pl.NB;              // buckets per job
  const size_t M = (size_t)k * jobstride;
  const int keystride = pl.Wb == 1 ? 0 : pl.NB;
  uint32_t* off = ws.off.as<uint32_t>();
  HeavyMeta* hm = ws.heavy_meta.as<HeavyMeta>();
  HeavyRec* hrecs = reinterpret_cast<HeavyRec*>(hm + 1);
  HIP_OK(hipMemsetAsync(hm, 0, sizeof(HeavyMeta), st));
  uint32_t* tiles = ws.scan_tmp.as<uint32_t>();
  {
    const int P = ceil_div((long)jobstride, 1L << PART_LOW_BITS);   // partitions per job
    const uint32_t pblk = batch.tile0[k];
    const size_t hn = (size_t)P * pblk;
    uint32_t* hist = ws.count.as<uint32_t>();
    uint32_t* hbase = hist + hn + 1;
    const int ht = ceil_div((long)hn + 1, 2048);
    uint32_t* total = tiles + ht;
    const uint32_t pgrid = pblk < PASS1_GRID ? pblk : PASS1_GRID;
    LAUNCH(k_part_hist, pgrid, 256, P * 4, st, batch, pl.c, pl.W, keystride, (int)scalars_mont, (int)fold, P, hist);
    // (a single-launch chained scan with decoupled look-back was measured in round 3 and is not kept: 0.080 ms against 0.052 ms for
    // these three launches on an empty chip, and no difference inside prove())
    LAUNCH(k_scan_tile_sums, ht, 256, 0, st, (const uint32_t*)hist, hn, tiles);
    LAUNCH(k_scan_top, 1, 256, 0, st, tiles, ht, total);
    LAUNCH(k_scan_apply, ht, 256, 0, st, (const uint32_t*)hist, hn, (const uint32_t*)tiles, hbase);
    // the LDS-staged passes need more dynamic LDS than the 64-KB default: asked for once per DEVICE (a function attribute belongs to the
    // device that is current when it is set; two threads racing here set the same values); a runtime that refuses keeps the direct
    // kernels instead of failing the MSM
    DeviceCtx& dctx = current_ctx();
    if (dctx.sort_staged < 0) {
      hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter_staged<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SORT_STAGE_MAX_LDS);
      if (e1 == hipSuccess) e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_scatter_staged<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SORT_STAGE_MAX_LDS);
      const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_part_sort), hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4);
      if (e1 != hipSuccess || e2 != hipSuccess) (void)hipGetLastError();
      dctx.sort_staged = (e1 == hipSuccess && e2 == hipSuccess) ? 1 : 0;
    }
    const bool staged_on = dctx.sort_staged == 1;
    const size_t stage_lds = (size_t)P * 8 + (size_t)PART_TILE * pl.W * 10;
    if (staged_on && stage_lds <= SORT_STAGE_MAX_LDS && P <= 65535) {
      static const bool planes = getenv("SONIC_SORT_PLANES") && atoi(getenv("SONIC_SORT_PLANES")) != 0;
      if (planes) LAUNCH(k_part_scatter_staged<true>, pgrid, PART_SCATTER_THREADS, stage_lds, st, batch, pl.c, pl.W, keystride, (int)scalars_mont, (int)fold, P, (const uint32_t*)hist,
                         (const uint32_t*)hbase, ws.digits.as<uint2>());
      else LAUNCH(k_part_scatter_staged<false>, pgrid, PART_SCATTER_THREADS, stage_lds, st, batch, pl.c, pl.W, keystride, (int)scalars_mont, (int)fold, P, (const uint32_t*)hist,
                  (const uint32_t*)hbase, ws.digits.as<uint2>());
    } else {
      LAUNCH(k_part_scatter, pgrid, 256, P * 4, st, batch, pl.c, pl.W, keystride, (int)scalars_mont, (int)fold, P, (const uint32_t*)hbase,
             ws.digits.as<uint2>());
    }
    const uint32_t nparts = (uint32_t)k * (uint32_t)P;
    // LDS stage of pass 2: room for 1.5x the mean partition of the largest job, a power of two between 2048 and 32768 entries (8 .. 128 KB)
    uint32_t stage_cap = 0;
    if (staged_on) {
      long nmax = 0;
      for (int j = 0; j < k; j++) nmax = jobs[j].n > nmax ? jobs[j].n : nmax;
      const long mean = nmax * pl.W / (P > 0 ? P : 1);
      stage_cap = 2048;
      while (stage_cap < 32768 && (long)stage_cap < mean + mean / 2) stage_cap *= 2;
    }
    LAUNCH(k_part_sort, nparts < PART_SORT_GRID ? nparts : PART_SORT_GRID, PART_SORT_THREADS, (size_t)stage_cap * 4, st, batch, (const uint2*)ws.digits.as<uint2>(),
           (const uint32_t*)hbase, (const uint32_t*)total, hn, P, jobstride, off, ws.entries.as<uint32_t>(), hm->class_hist, stage_cap);
  }
  LAUNCH(k_border_place, ceil_div((long)M, 2048), 256, 0, st, (const uint32_t*)off, (uint32_t)M, (const uint32_t*)hm->class_hist, hm->class_cursor,
         ws.order.as<uint32_t>());
  const int accum_block = (pl.accum_block == 64 || pl.accum_block == 128) ? pl.accum_block : 256;
  // lanes per bucket: enough of them for two rounds of the chip's 2048 wave slots (SONIC_ACCUM_LANES=1: always one)
  static const int split_env = getenv("SONIC_ACCUM_LANES") ? atoi(getenv("SONIC_ACCUM_LANES")) : 0;
  const int lanes_per_bucket = split_env == 1 ? 1 : (M <= 65536 ? 4 : (M <= 131072 ? 2 : 1));
  if (lanes_per_bucket == 4)
    LAUNCH(k_bucket_accum_split<4>, ceil_div(4 * (long)M, 256), 256, 0, st, batch, jobstride, (const uint32_t*)ws.entries.as<uint32_t>(),
           (const uint32_t*)off, (const uint32_t*)ws.order.as<uint32_t>(), (uint32_t)M, pl.heavy_threshold, buckets, hm, hrecs);
  else if (lanes_per_bucket == 2)
    LAUNCH(k_bucket_accum_split<2>, ceil_div(2 * (long)M, 256), 256, 0, st, batch, jobstride, (const uint32_t*)ws.entries.as<uint32_t>(),
           (const uint32_t*)off, (const uint32_t*)ws.order.as<uint32_t>(), (uint32_t)M, pl.heavy_threshold, buckets, hm, hrecs);
  else
  LAUNCH(k_bucket_accum, ceil_div((long)M, accum_block), accum_block, 0, st, batch, jobstride, (const uint32_t*)ws.entries.as<uint32_t>(),
         (const uint32_t*)off, (const uint32_t*)ws.order.as<uint32_t>(), (uint32_t)M, pl.heavy_threshold, buckets, hm, hrecs);
  LAUNCH(k_heavy_accum, HEAVY_GRID, HEAVY_THREADS, 0, st, batch, jobstride, (const uint32_t*)ws.entries.as<uint32_t>(), (const uint32_t*)off,
         (const HeavyMeta*)hm, hrecs, ws.heavy_partial.as<G1XYZZ>());
  LAUNCH(k_heavy_finish, HEAVY_GRID / 2, 64, 0, st, (const HeavyMeta*)hm, (const HeavyRec*)hrecs,
         (const G1XYZZ*)ws.heavy_partial.as<G1XYZZ>(), buckets);
  if (ext_buckets) return;                   // the caller reduces the buckets (msm_reduce_slices_enqueue, possibly on another rank)
  const int sets = k * pl.Wb;
  if ((pl.tree && pl.Wb == 1 && pl.NB >= 4) || pl.endo) {
    int L = 0;
    while ((1 << L) < pl.NB) L++;
    bucket_tree_enqueue(st, ws.buckets.as<G1XYZZ>(), sets, L, 1u, batch);
    return;
  }
  LAUNCH(k_bucket_segments, ceil_div((long)sets * pl.nseg, 256), 256, 0, st, (const G1XYZZ*)ws.buckets.as<G1XYZZ>(), sets,
         pl.NB, pl.K, pl.nseg, 0, ws.segres.as<G1XYZZ>());
  if (pl.nseg > 4096) {
    // sets with tens of thousands of segments: 256-way groups first (nseg is a power of two), then one tree per set
    const int group = 256, ngroups = pl.nseg / group;
    G1XYZZ* part = ws.segres.as<G1XYZZ>() + (size_t)sets * pl.nseg;
    LAUNCH(k_group_sum, sets * ngroups, 256, 0, st, (const G1XYZZ*)ws.segres.as<G1XYZZ>(), (long)sets * pl.nseg, group, part);
    LAUNCH(k_window_sum, sets, 256, 0, st, (const G1XYZZ*)part, pl.Wb, pl.c, ngroups, batch);
  } else {
    LAUNCH(k_window_sum, sets, 256, 0, st, (const G1XYZZ*)ws.segres.as<G1XYZZ>(), pl.Wb, pl.c, pl.nseg, batch);
  }
}

// ---- one rank's share of an MSM whose BUCKETS are sharded across ranks ---------------------------------------------------
// Every rank accumulates its term range into a full bucket set (msm_enqueue_batch with ext_buckets), the ranks exchange bucket
// ranges (all-to-all), and each rank then owns `k` slices of ONE range of `len` buckets starting at bucket `base`:
//   sum_{i < len} (base + i + 1) * (sum_s slices[s][i])
// -- the element-wise curve addition of what the ranks sent, then the usual running sums with the range's weights.
__global__ __launch_bounds__(256, 1) void k_sum_slices(const G1XYZZ* __restrict__ slices, int k, long len, G1XYZZ* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  G1XYZZ acc = slices[i];
  for (int s = 1; s < k; s++) acc = g1_add(acc, slices[(size_t)s * len + i]);
  out[i] = acc;
}

void msm_reduce_slices_enqueue(hipStream_t st, MsmWorkspace& ws, const G1XYZZ* d_slices, int k, long len, long base, int c, MsmSlot* d_slot) {
  (void)c;
  if (k < 1 || len < 4 || len % MSM_SLICE_QUANTUM || base % MSM_SLICE_QUANTUM || base + len >= (1L << 31))
    throw std::runtime_error("msm_reduce_slices_enqueue: slice length and base must be multiples of MSM_SLICE_QUANTUM");
  // the butterfly wants a power of two: the sum of the slices goes into a zero-padded (= infinity) copy; bucket i weighs base + i + 1
  // (up to round 3 this was a running sum over 2-bucket segments: ~1.0 ms of dependent chain for a 1/8 slice, 0.29 ms as a tree)
  int L = 2;
  while (((long)1 << L) < len) L++;
  const long P = (long)1 << L;
  ws.buckets.ensure((size_t)P * sizeof(G1XYZZ));
  if (P > len) HIP_OK(hipMemsetAsync(ws.buckets.as<G1XYZZ>() + len, 0, (size_t)(P - len) * sizeof(G1XYZZ), st));
  // (a quad per bucket, and four lanes per bucket with an LDS tree, both measured slower at 65536 buckets x 8 slices -- 0.21 / 0.22 against
  // 0.15 ms: the kernel reads 100 MB once and is bound by that, not by its chain of 7 additions)
  // (round 6: two and four lanes per bucket with the partial sums folded over DPP -- 4 and 3 additions deep instead of 7 -- measured 0.146 and
  // 0.211 ms against 0.145-0.149: with every SIMD holding a wave the kernel is bound by the additions it issues, 7 per bucket at the least;
  // profiles/r06_msm_strong.txt)
  LAUNCH(k_sum_slices, ceil_div(len, 256), 256, 0, st, d_slices, k, len, ws.buckets.as<G1XYZZ>());
  MsmBatchDev b1;
  memset(&b1, 0, sizeof b1);
  b1.k = 1;
  b1.bits = 255;
  b1.slot_form = 1;
  b1.slot[0] = d_slot;
  bucket_tree_enqueue(st, ws.buckets.as<G1XYZZ>(), 1, L, (uint32_t)(base + 1), b1);
}

void msm_enqueue(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, PointArray d_points, const Fr* d_scalars,
                 long n, bool scalars_mont, MsmSlot* d_slot) {
  MsmJob job{d_points, d_scalars, n, d_slot};
  msm_enqueue_batch(st, ws, pl, &job, 1, scalars_mont);
}

}  // namespace sonic
