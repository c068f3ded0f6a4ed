// C ABI of libsonic_hip.so (include/sonic_hip.h): library state, encodings at the boundary,
// the SRS handle and the standalone MSM / NTT entry points.  prove lives in prove.hip.
#include <stdarg.h>
#include <string.h>
#include <condition_variable>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "internal.hpp"
#include "endo.hpp"
#include "g2.hpp"
#include "srs_file.hpp"

namespace sonic {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

Profiler& profiler() { static Profiler p; return p; }
void Profiler::collect() {
  std::lock_guard<std::mutex> g(mu);
  for (auto& kv : recs) {
    auto& tot = totals[kv.first];
    for (auto& r : kv.second) {
      hipEventSynchronize(r.b);
      float ms = 0;
      hipEventElapsedTime(&ms, r.a, r.b);
      tot.first += ms; tot.second += 1;
      hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    kv.second.clear();
  }
}
void Profiler::reset() { collect(); std::lock_guard<std::mutex> g(mu); totals.clear(); }

// ---- devices ----------------------------------------------------------------------------------------------------------------------
static std::mutex g_init_mu;
static int g_device = -1;                    // the default device: sonic_init, else LOCAL_RANK % count, else 0
static int g_device_count = -1;
static std::vector<DeviceCtx*> g_ctx;        // by ordinal; entries are made on first use and live as long as the process
static thread_local DeviceCtx* t_ctx = nullptr;

static int device_count_locked() {
  if (g_device_count >= 0) return g_device_count;
  // one hardware queue per prover stream (the runtime's default of 4 makes streams queue behind each other); only effective when
  // this is the process's first HIP call, harmless otherwise
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (%s): libsonic_hip has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    throw HipFail{SONIC_ERR_NO_DEVICE};
  }
  g_device_count = n;
  g_ctx.assign((size_t)n, nullptr);
  return n;
}
// caller holds g_init_mu and has made `dev` the thread's HIP device
static DeviceCtx* ctx_locked(int dev) {
  if (!g_ctx[(size_t)dev]) {
    std::unique_ptr<DeviceCtx> c(new DeviceCtx());
    c->dev = dev;
    HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    g_ctx[(size_t)dev] = c.release();
  }
  return g_ctx[(size_t)dev];
}
int default_device_ordinal() { std::lock_guard<std::mutex> g(g_init_mu); return g_device; }
void unlink_one_shot_of(const sonic_srs* s) {
  std::lock_guard<std::mutex> g(g_init_mu);
  for (DeviceCtx* c : g_ctx) {
    if (!c) continue;
    std::lock_guard<std::mutex> g2(c->one_shot_mu);
    for (size_t i = 0; i < c->one_shot.size();) {
      if (static_cast<OneShotShell*>(c->one_shot[i])->srs == s) c->one_shot.erase(c->one_shot.begin() + (long)i);      // (leaked: its device is out of reach)
      else i++;
    }
  }
}

DeviceScope::DeviceScope(int dev) : ctx_(nullptr), prev_ctx_(t_ctx), prev_dev_(-1) {
  if (t_ctx && (dev < 0 || t_ctx->dev == dev)) { ctx_ = t_ctx; return; }      // nested call: a handle-less callee inherits the caller's device
  std::lock_guard<std::mutex> g(g_init_mu);
  const int n = device_count_locked();
  if (dev < 0) {
    if (g_device < 0) {
      const char* lr = getenv("LOCAL_RANK");
      g_device = lr ? atoi(lr) % n : 0;
      if (g_device < 0) g_device = 0;
    }
    dev = g_device;
  }
  if (dev >= n) { set_error("device %d out of range (%d device%s)", dev, n, n == 1 ? "" : "s"); throw HipFail{SONIC_ERR_INVALID_ARG}; }
  if (hipGetDevice(&prev_dev_) != hipSuccess) { (void)hipGetLastError(); prev_dev_ = -1; }
  if (prev_dev_ != dev) HIP_OK(hipSetDevice(dev));
  try { ctx_ = ctx_locked(dev); } catch (...) { if (prev_dev_ >= 0 && prev_dev_ != dev) (void)hipSetDevice(prev_dev_); throw; }
  t_ctx = ctx_;
}
DeviceScope::~DeviceScope() {
  if (ctx_ == prev_ctx_) return;                       // nested on the same device: nothing was changed
  t_ctx = prev_ctx_;
  if (prev_dev_ >= 0 && prev_dev_ != ctx_->dev) (void)hipSetDevice(prev_dev_);
}
const NttTables& device_ntt_tables(int log2n) {
  DeviceCtx& c = current_ctx();
  std::lock_guard<std::mutex> g(c.pool_mu);
  NttTables*& t = c.prover_ntt[log2n];
  if (!t) {
    std::unique_ptr<NttTables> nt(new NttTables());
    hipStream_t st = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    try { nt->ensure(st, log2n); HIP_OK(hipStreamSynchronize(st)); } catch (...) { (void)hipStreamDestroy(st); throw; }
    (void)hipStreamDestroy(st);
    t = nt.release();
  }
  return *t;
}
DeviceCtx& current_ctx() {
  if (!t_ctx) { set_error("internal: no device scope on this thread"); throw HipFail{SONIC_ERR_HIP}; }
  return *t_ctx;
}

CallLease::CallLease() : c_(nullptr), owner_(&current_ctx()) {
  {
    std::lock_guard<std::mutex> g(owner_->pool_mu);
    if (!owner_->pool.empty()) { c_ = owner_->pool.back(); owner_->pool.pop_back(); }
  }
  if (!c_) {
    std::unique_ptr<CallCtx> c(new CallCtx());
    HIP_OK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    c_ = c.release();
  }
}
CallLease::~CallLease() {
  (void)hipStreamSynchronize(c_->st);        // nothing of this call may still be running when the context is handed on
  std::lock_guard<std::mutex> g(owner_->pool_mu);
  owner_->pool.push_back(c_);
}

// ---- encodings ------------------------------------------------------------------------------
// inf_ok: index at which the point at infinity is accepted (-1: everywhere, as for the operands of sonic_msm_g1; -2: nowhere; an SRS has
// exactly one such slot, the omitted g^alpha); elsewhere infinity sets err bit 8 -- g^{x^e} and g^{alpha x^e} are never the
// identity for x, alpha != 0, and a zero-filled SRS must not pass for a valid one
__global__ __launch_bounds__(256) void k_points_from_bytes(const uint8_t* __restrict__ in, PointArrayMut out, long n, int* err, long inf_ok) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(in + 96 * i);
  G1Affine p;
  for (int k = 0; k < 12; k++) { p.x.l[k] = w[k]; p.y.l[k] = w[12 + k]; }
  if (!fp_is_canonical(p.x) || !fp_is_canonical(p.y)) { atomicOr(err, 1); out[i] = G1Affine::inf(); return; }   // before is_inf: (q, q) is not O
  if (p.is_inf()) { if (inf_ok != -1 && i != inf_ok) atomicOr(err, 8); out[i] = p; return; }
  p.x = fp_to_mont(p.x); p.y = fp_to_mont(p.y);
  Fq four = fp_dbl(fp_dbl(Fq::one()));
  if (fp_sqr(p.y) != fp_add(fp_mul(fp_sqr(p.x), p.x), four)) { atomicOr(err, 2); out[i] = G1Affine::inf(); return; }
  out[i] = p;
}
// r P == O for every point (literal double-and-add over the bits of r): SRS elements must lie in the order-r subgroup because
// MSMs over an SRS fold scalars with r P = O (msm.hpp).  Sets err bit 4 otherwise.
__global__ __launch_bounds__(256) void k_points_subgroup_check(PointArray in, long n, int* err) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const G1Affine p = in[i];
  if (p.is_inf()) return;
  constexpr uint32_t rl[8] = FR_P;
  G1XYZZ acc = G1XYZZ::from_affine(p);       // top bit (254) of r
#pragma unroll 1
  for (int b = 253; b >= 0; b--) {
    acc = g1_dbl(acc);
    uint32_t w = 0;                            // constant-index reads keep rl[] out of scratch
#pragma unroll
    for (int k = 0; k < 8; k++) if (k == (b >> 5)) w = rl[k];
    if ((w >> (b & 31)) & 1u) acc = g1_add_mixed(acc, p);
  }
  if (!acc.is_inf()) atomicOr(err, 4);
}
__global__ __launch_bounds__(256) void k_points_to_bytes(PointArray in, uint8_t* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  G1Affine p = in[i];
  uint32_t* w = reinterpret_cast<uint32_t*>(out + 96 * i);
  if (p.is_inf()) { for (int k = 0; k < 24; k++) w[k] = 0; return; }
  Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
  for (int k = 0; k < 12; k++) { w[k] = x.l[k]; w[12 + k] = y.l[k]; }
}
__global__ __launch_bounds__(256) void k_fr_check(const Fr* __restrict__ in, long n, int* err) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr a = in[i];
  if (!fp_is_canonical(a)) atomicOr(err, 1);
}
__global__ __launch_bounds__(256) void k_fr_to_mont(Fr* __restrict__ a, long n, int* err) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr v = a[i];
  if (!fp_is_canonical(v)) { atomicOr(err, 1); return; }
  a[i] = fp_to_mont(v);
}
__global__ __launch_bounds__(256) void k_fr_from_mont(Fr* __restrict__ a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = fp_from_mont(a[i]);
}

void fr_to_mont_enqueue(hipStream_t st, Fr* d, long n, int* d_err) { if (n > 0) LAUNCH(k_fr_to_mont, ceil_div(n, 256), 256, 0, st, d, n, d_err); }
void fr_from_mont_enqueue(hipStream_t st, Fr* d, long n) { if (n > 0) LAUNCH(k_fr_from_mont, ceil_div(n, 256), 256, 0, st, d, n); }
void fr_check_enqueue(hipStream_t st, const Fr* d, long n, int* d_err) { if (n > 0) LAUNCH(k_fr_check, ceil_div(n, 256), 256, 0, st, d, n, d_err); }

// one MSM, finished and normalised, result on the host
void msm_blocking(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, PointArray d_pts, const Fr* d_sc, long n, bool mont,
                  uint8_t* out96, uint8_t* out_partial192) {
  DevBuf slot(sizeof(MsmSlot));
  msm_enqueue(st, ws, pl, d_pts, d_sc, n, mont, slot.as<MsmSlot>());
  MsmSlot h;
  HIP_OK(hipMemcpyAsync(&h, slot.p, sizeof(MsmSlot), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  G1XYZZ sum = msm_finish_host(h);
  if (out96) g1_canonical_bytes_host(sum, out96);
  if (out_partial192) memcpy(out_partial192, &sum, sizeof sum);
}

}  // namespace sonic

using namespace sonic;

// every entry point runs inside a DeviceScope: API_BEGIN on the default device, API_BEGIN_ON(dev) on a handle's device
#define API_BEGIN_ON(dev) try { ::sonic::DeviceScope _scope(dev);
#define API_BEGIN API_BEGIN_ON(-1)
#define API_END                                                        \
  } catch (const HipFail& f) { return f.code; }                        \
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; } \
  return SONIC_OK;

struct sonic_srs {
  int64_t d;
  int device = 0;             // the GPU that holds the handle's memory: every call that takes the handle runs there (DeviceScope)
  // Fiat-Shamir id of the reference string (fs.hpp): constant for an SRS, made on first use (four point fetches) and cached
  mutable std::mutex id_mu;
  mutable bool have_id = false;
  mutable uint8_t id[32] = {0};
  // basis b, window table w, exponent e  ->  tab[b][w * (2d+1) + e + d] = 2^(msm_even_shift(tab_W, w)) * g^{(alpha^b) x^e}
  // (w = 0 is the basis itself; tab_W = 1 when the window tables are switched off)
  int tab_c = 0, tab_W = 1;
  // endomorphism tables (endo.hpp): tab_W windows over 130 bits instead of 255 -- 7 tables instead of 13 at c = 19 / 20 -- for SRS
  // sizes whose full tables do not fit; every MSM then runs as two half-scalar MSMs (msm_enqueue_batch)
  bool tab_endo = false;
  DevBuf g, ga;
  DevBuf ps;                 // running sums of the alpha basis (srs.hip, srs_build_prefix); empty when memory was short
  DevBuf gs;                 // symmetric sums A[e] + A[-e] of the alpha basis with their window tables (srs_build_sym); empty when memory was short
  // verifier half: generated on first use from the trapdoor SRS.new was given -- which is wiped as soon as that has
  // happened -- or attached by sonic_srs_set_g2_points / read from a version-2 file
  mutable bool have_trapdoor = false;
  mutable Fr x_std, alpha_std;
  mutable std::mutex g2_mu;
  mutable DevBuf h, ha;      // G2Affine[2d+1] each
  ~sonic_srs() { wipe_trapdoor(); }
  void wipe_trapdoor() const {
    volatile uint32_t* a = x_std.l; volatile uint32_t* b = alpha_std.l;
    for (int i = 0; i < 8; i++) { a[i] = 0; b[i] = 0; }
    have_trapdoor = false;
  }
  PointArray basis(int b) const { return PointArray{(b ? ga : g).as<char>(), SONIC_SRS_POINT_BYTES}; }
};

namespace sonic {
PointArray srs_basis(const sonic_srs* s, int b) { return s->basis(b); }
int64_t srs_d(const sonic_srs* s) { return s->d; }
int srs_device(const sonic_srs* s) { return s ? s->device : -1; }
int srs_cached_id(const sonic_srs* s, int (*make)(const sonic_srs*, uint8_t*), uint8_t out[32]) {
  // only a success is kept (a transient failure of the first attempt -- no memory for the leased context, a busy device --
  // used to be remembered with its status, and sonic_prover_prove_fs / sonic_verify_fs failed on that handle from then on although its
  // points were fine): a failed attempt is simply made again by the next call
  std::lock_guard<std::mutex> g(s->id_mu);
  if (!s->have_id) {
    uint8_t tmp[32];
    const int rc = make(s, tmp);
    if (rc) return rc;
    memcpy(s->id, tmp, 32);
    s->have_id = true;
  }
  memcpy(out, s->id, 32);
  return SONIC_OK;
}
int srs_tab_c(const sonic_srs* s) { return s->tab_c; }
int srs_tab_W(const sonic_srs* s) { return s->tab_W; }
bool srs_tab_endo(const sonic_srs* s) { return s->tab_endo; }

// Window tables trade HBM capacity (288 GB) for work: W x the SRS size buys one shared bucket set per MSM.
// c grows with d (MSM sizes are a fraction of d); off with SONIC_MSM_TABLES=0 or when memory is short.
sonic_srs* srs_alloc(int64_t d) {
  sonic_srs* s = new sonic_srs();
  s->d = d;
  s->device = current_ctx().dev;
  const size_t n = (size_t)(2 * d + 1);
  int lg = 0;
  while ((2L << lg) <= d) lg++;                 // floor(log2 d)
  // measured (prove at n = d/8, profiles/r05_table_c_ab.txt): up to d = 2^19 the MSMs (0.4 d .. 0.9 d terms) run best with ~2^16
  // bucket walks (c = 17: one to two waves per SIMD, short reduction); from d = 2^20 the two windows saved by c = 20 win (prove at
  // n = 2^17 20.2 -> 19.2 ms; a stand-alone MSM of 0.9 d terms 5.7 -> 2.9 ms: 2^16 walks of 210 entries are one wave per SIMD).
  int c = lg >= 20 ? 20 : (lg > 17 ? 17 : lg);
  // round 6, d = 2^16, 2^17: c = 16.  A proof's MSMs over such an SRS run as ONE chain (prove.hip, fused) whose accumulation is 45 n W
  // additions and whose butterfly costs ~2.8 additions' worth per bucket of 15 bucket sets: at n = d/8 = 2^14 the 2^16-bucket plan (c = 17)
  // spent 0.82 ms reducing beside 1.75 ms accumulating (profiles/r06_small_proofs.txt).  Measured at n = 2^14, ms per proof streamed / one
  // at a time: c = 17 4.09-4.18 / 4.27-4.54, c = 16 3.91-3.98 / 4.23-4.33, c = 15 4.09-4.11 / 4.44-4.51, c = 14 8.5 / 9.0 (15 x 2^13 bucket
  // walks are less than one round of the chip's wave slots and 350-800 entries long); at n = 2^16 c = 16 11.1 against 10.5-10.7 at 17
  // (profiles/r06_ab_small.txt).
  if (lg >= 15 && lg <= 17) c = lg == 15 ? 15 : 16;
  if (c < 9) c = 9;
  if (const char* tc = getenv("SONIC_MSM_TABLE_C")) { const int v = atoi(tc); if (v >= 9 && v <= 22) c = v; }
  // W windows of even width (msm.hpp): the widest is ceil(255 / W) <= c
  int W = (255 + c - 1) / c;
  const int c_full = (255 + W - 1) / W;
  const char* env = getenv("SONIC_MSM_TABLES");
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const size_t per_table = 2 * n * (size_t)SONIC_SRS_POINT_BYTES;
  // endomorphism tables: windows over 130 bits -- half as many -- when the full set does not fit half of the free memory (or on
  // request: SONIC_MSM_ENDO=1, tests); one more addition per term and window-pair than the full tables, no tables at all costs 2.5x
  const int W_endo = (ENDO_BITS + c - 1) / c;
  const char* eenv = getenv("SONIC_MSM_ENDO");
  bool endo = false;
  const bool fits_full = per_table * (size_t)W <= free_b / 2, fits_endo = per_table * (size_t)W_endo <= free_b / 2;
  const bool endo_forced = eenv && atoi(eenv) == 1, endo_off = eenv && atoi(eenv) == 0;
  if (env && atoi(env) == 0) { c = 0; W = 1; }
  else if ((endo_forced || (!fits_full && !endo_off)) && fits_endo) { endo = true; W = W_endo; c = (ENDO_BITS + W - 1) / W; }
  else if (fits_full && !endo_forced) c = c_full;
  else { c = 0; W = 1; }
  s->tab_c = c; s->tab_W = W; s->tab_endo = endo;
  s->g.alloc((size_t)SONIC_SRS_POINT_BYTES * n * W);
  s->ga.alloc((size_t)SONIC_SRS_POINT_BYTES * n * W);
  // the running sums of the alpha basis: one more table, where 1/13 of what the window tables took is still to be had
  (void)hipMemGetInfo(&free_b, &total_b);
  const char* penv = getenv("SONIC_SRS_PREFIX");
  if (!(penv && atoi(penv) == 0) && (size_t)SONIC_SRS_POINT_BYTES * n <= free_b / 4) s->ps.alloc((size_t)SONIC_SRS_POINT_BYTES * n);
  // the symmetric sums of the alpha basis with window tables of their own: half of what the two bases took, where a quarter of the rest holds it
  // (only with the full tables: the job over them shares a batched chain with jobs over the bases)
  (void)hipMemGetInfo(&free_b, &total_b);
  const char* senv = getenv("SONIC_SRS_SYM");
  // (round 6: d + 1 slots per window -- exponents 0 .. d -- instead of a whole basis of 2d + 1)
  const size_t sym_bytes = (size_t)SONIC_SRS_POINT_BYTES * (size_t)(d + 1) * W;
  if (!(senv && atoi(senv) == 0) && W > 1 && !endo && sym_bytes <= free_b / 4) s->gs.alloc(sym_bytes);
  return s;
}
PointArrayMut srs_sym_mut(sonic_srs* s) { return PointArrayMut{s->gs.as<char>(), SONIC_SRS_POINT_BYTES}; }
PointArray srs_sym(const sonic_srs* s) { return PointArray{s->gs.as<char>(), SONIC_SRS_POINT_BYTES}; }
PointArrayMut srs_prefix_mut(sonic_srs* s) { return PointArrayMut{s->ps.as<char>(), SONIC_SRS_POINT_BYTES}; }
PointArray srs_prefix(const sonic_srs* s) { return PointArray{s->ps.as<char>(), SONIC_SRS_POINT_BYTES}; }
PointArrayMut srs_basis_mut(sonic_srs* s, int b) { return PointArrayMut{(b ? s->ga : s->g).as<char>(), SONIC_SRS_POINT_BYTES}; }
void srs_set_trapdoor(sonic_srs* s, const Fr& x_std, const Fr& alpha_std) { s->have_trapdoor = true; s->x_std = x_std; s->alpha_std = alpha_std; }

// plan for an MSM over n consecutive SRS points: shared-bucket plan over the window tables unless the MSM is
// tiny compared with the bucket set, the tables are off, or a test forces a window size
// Entry encoding limits (msm.hip, k_part_scatter / entry_point): over window tables an entry packs the term index into 26 bits and
// the window into 5, without tables the index has 31 bits.  MSMs of 2^26 terms and more therefore run over per-window buckets
// even when the SRS has tables, and 2^31 terms are refused (msm_enqueue_batch, sonic_msm_plan).
MsmPlan srs_msm_plan(const sonic_srs* s, long n) {
  // (up to round 5 an MSM of fewer than 1/16 of the bucket count fell back to per-window buckets without the tables: 64 four-bit windows,
  // 64 bucket sets, a chain that cannot be batched and 255 doublings of host tail per MSM -- 0.9 ms each, which is what the reference's
  // own criterion shape, n = 1 / d = 25, paid fifteen times per proof: profiles/r05_criterion_shape.txt.  Over the tables the same MSM
  // is a batchable job whose cost is the butterfly over the shared bucket set: 0.12 ms at 2^16 buckets, 0.55 ms at 2^19.)
  if (s->tab_W > 1 && s->tab_W <= MSM_TABLE_MAX_WINDOWS && n < MSM_TABLE_MAX_TERMS && msm_window_override() == 0)
    return msm_plan_tables(n > 0 ? n : 1, s->tab_c, s->tab_W, 2 * s->d + 1, s->tab_endo);
  return msm_plan(n > 0 ? n : 1);
}
}  // namespace sonic

extern "C" {

int sonic_init(int device_ordinal) {
  try {
    {
      std::lock_guard<std::mutex> g(g_init_mu);
      const int n = device_count_locked();
      if (device_ordinal >= n) { set_error("device %d out of range (%d device%s)", device_ordinal, n, n == 1 ? "" : "s"); return SONIC_ERR_INVALID_ARG; }
      if (g_device < 0 && device_ordinal >= 0) g_device = device_ordinal;       // the first choice of a default device stands
    }
    int dev;
    { DeviceScope scope(-1); dev = scope.ctx().dev; }
    // the default device stays the calling thread's HIP device after sonic_init (a caller with a HIP binding of its own, e.g. torch
    // tensors handed to the _dev entry points, allocates there)
    HIP_OK(hipSetDevice(dev));
  } catch (const HipFail& f) { return f.code; }
  return SONIC_OK;
}

int sonic_device_count(int* out) {
  if (!out) return SONIC_ERR_INVALID_ARG;
  try { std::lock_guard<std::mutex> g(g_init_mu); *out = device_count_locked(); } catch (const HipFail& f) { *out = 0; return f.code; }
  return SONIC_OK;
}

// the HIP the library was built against and the HIP runtime that got mapped into this process (they differ when another
// component, e.g. a PyTorch-ROCm wheel, brought its own libamdhip64 first); no device needed
int sonic_hip_versions(int* build, int* runtime) {
  if (build) *build = HIP_VERSION;
  if (runtime) { int v = 0; if (hipRuntimeGetVersion(&v) != hipSuccess) v = 0; *runtime = v; }
  return SONIC_OK;
}

int sonic_last_error(char* buf, size_t cap) {
  if (!buf || cap == 0) return SONIC_ERR_INVALID_ARG;
  strncpy(buf, g_err, cap - 1);
  buf[cap - 1] = 0;
  return SONIC_OK;
}

int sonic_device_sync(void) { API_BEGIN HIP_OK(hipStreamSynchronize(default_stream())); HIP_OK(hipDeviceSynchronize()); API_END }

int sonic_srs_from_points(int64_t d, const uint8_t* basis0, const uint8_t* basis1, sonic_srs_t** out) {
  return sonic_srs_from_points_on(-1, d, basis0, basis1, out);
}
int sonic_srs_from_points_on(int device, int64_t d, const uint8_t* basis0, const uint8_t* basis1, sonic_srs_t** out) {
  API_BEGIN_ON(device)
  if (d < 1 || !basis0 || !basis1 || !out) { set_error("sonic_srs_from_points: bad argument"); return SONIC_ERR_INVALID_ARG; }
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  const long n = 2 * d + 1;
  sonic_srs* s = srs_alloc(d);
  DevBuf raw(96 * n), err(4);
  HIP_OK(hipMemsetAsync(err.p, 0, 4, st));
  for (int b = 0; b < 2; b++) {
    HIP_OK(hipMemcpyAsync(raw.p, b ? basis1 : basis0, 96 * n, hipMemcpyHostToDevice, st));
    LAUNCH(k_points_from_bytes, ceil_div(n, 256), 256, 0, st, (const uint8_t*)raw.as<uint8_t>(), srs_basis_mut(s, b), n, err.as<int>(),
           b ? (long)d : -2L);        // basis 1 has the empty slot e = 0 (SRS.hs:38); basis 0 has none
    LAUNCH(k_points_subgroup_check, ceil_div(n, 256), 256, 0, st, (PointArray)srs_basis_mut(s, b), n, err.as<int>());
  }
  int herr = 0;
  HIP_OK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (herr) {
    delete s;
    set_error("sonic_srs_from_points: %s", (herr & 1) ? "non-canonical coordinate" : (herr & 2) ? "point not on curve" : (herr & 4) ? "point outside the order-r subgroup" : "point at infinity (only the omitted g^alpha, basis 1 slot e = 0, may be empty)");
    return SONIC_ERR_BAD_ENCODING;
  }
  srs_build_tables(st, s);
  *out = s;
  API_END
}

void sonic_srs_free(sonic_srs_t* srs) {
  if (!srs) return;
  // (when the handle's device cannot be made current the parked one-shot shells over this SRS must still be unlinked -- a later
  // SRS allocated at the same address would otherwise match a shell whose tables are gone)
  try { DeviceScope scope(srs->device); drop_one_shot_of(srs); delete srs; } catch (const HipFail&) { unlink_one_shot_of(srs); delete srs; }
}
int64_t sonic_srs_d(const sonic_srs_t* srs) { return srs ? srs->d : -1; }
int sonic_srs_device(const sonic_srs_t* srs) { return srs ? srs->device : -1; }

int sonic_srs_get_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || !out || n < 0) return SONIC_ERR_INVALID_ARG;
  if (e0 < -srs->d || e0 + n - 1 > srs->d) { set_error("sonic_srs_get_points: exponent range [%ld, %ld] outside [-%ld, %ld]", (long)e0, (long)(e0 + n - 1), (long)srs->d, (long)srs->d); return SONIC_ERR_SRS_INDEX; }
  if (n == 0) return SONIC_OK;
  CallLease lease;
  hipStream_t st = lease.st();
  DevBuf raw(96 * n);
  if (basis == SONIC_BASIS_ALPHA_SYM) {        // diagnostic: the symmetric sums of the alpha basis (srs_build_sym; entries e <= 0 are empty)
    if (!srs->gs.p) { set_error("sonic_srs_get_points: this SRS holds no symmetric sums"); return SONIC_ERR_INVALID_ARG; }
    // (the table holds exponents 0 .. d: slot 0 and everything below it read as the empty sum)
    HIP_OK(hipMemsetAsync(raw.p, 0, 96 * n, st));
    const int64_t lo = e0 < 0 ? 0 : e0, hi = e0 + n;              // exponents [lo, hi) are in the table
    if (hi > lo) LAUNCH(k_points_to_bytes, ceil_div(hi - lo, 256), 256, 0, st, srs_sym(srs) + lo, raw.as<uint8_t>() + 96 * (lo - e0), (long)(hi - lo));
    HIP_OK(hipMemcpyAsync(out, raw.p, 96 * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return SONIC_OK;
  }
  if (basis == SONIC_BASIS_ALPHA_PREFIX) {     // diagnostic: the running sums of the alpha basis (srs_build_prefix)
    if (!srs->ps.p) { set_error("sonic_srs_get_points: this SRS holds no running sums"); return SONIC_ERR_INVALID_ARG; }
    LAUNCH(k_points_to_bytes, ceil_div(n, 256), 256, 0, st, srs_prefix(srs) + (e0 + srs->d), raw.as<uint8_t>(), (long)n);
    HIP_OK(hipMemcpyAsync(out, raw.p, 96 * n, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    return SONIC_OK;
  }
  // diagnostic: basis = b + 2 w reads window table w (2^(c w) multiples) of basis b
  const int w = basis >> 1;
  if (w >= srs->tab_W) { set_error("sonic_srs_get_points: no window table %d", w); return SONIC_ERR_INVALID_ARG; }
  LAUNCH(k_points_to_bytes, ceil_div(n, 256), 256, 0, st, srs->basis(basis & 1) + (size_t)w * (2 * srs->d + 1) + (e0 + srs->d), raw.as<uint8_t>(), (long)n);
  HIP_OK(hipMemcpyAsync(out, raw.p, 96 * n, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  API_END
}

// makes sure the G2 half is resident; caller holds call_mutex
static int srs_ensure_g2(const sonic_srs_t* srs, hipStream_t st, const char* who) {
  std::lock_guard<std::mutex> g2(srs->g2_mu);
  if (srs->h.p) return SONIC_OK;
  if (!srs->have_trapdoor) { set_error("%s: this SRS has no G2 half (built from G1 points only)", who); return SONIC_ERR_INVALID_ARG; }
  const size_t cnt = (size_t)(2 * srs->d + 1);
  // generated into local buffers and attached only when the generation has finished: a failure half way (e.g. no memory for
  // its Jacobian scratch at large d) must leave the handle without a G2 half and with its trapdoor, not with uninitialised
  // memory that a later call would serve -- all-zero bytes decode as the point at infinity, and a verifier whose three G2
  // elements are at infinity accepts every proof
  DevBuf h0(sizeof(G2Affine) * cnt), h1(sizeof(G2Affine) * cnt);
  srs_generate_g2(st, srs->d, srs->x_std, srs->alpha_std, h0.as<G2Affine>(), h1.as<G2Affine>());
  srs->h = std::move(h0);
  srs->ha = std::move(h1);
  srs->wipe_trapdoor();          // x and alpha have served their last purpose
  return SONIC_OK;
}

int sonic_srs_get_g2_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || !out || n < 0 || (basis != 0 && basis != 1)) return SONIC_ERR_INVALID_ARG;
  if (e0 < -srs->d || e0 + n - 1 > srs->d) { set_error("sonic_srs_get_g2_points: exponent range [%ld, %ld] outside [-%ld, %ld]", (long)e0, (long)(e0 + n - 1), (long)srs->d, (long)srs->d); return SONIC_ERR_SRS_INDEX; }
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  int rc = srs_ensure_g2(srs, st, "sonic_srs_get_g2_points");
  if (rc) return rc;
  if (n == 0) return SONIC_OK;
  DevBuf raw(192 * n);
  g2_points_to_bytes_enqueue(st, (basis ? srs->ha : srs->h).as<G2Affine>() + (e0 + srs->d), raw.as<uint8_t>(), n);
  HIP_OK(hipMemcpyAsync(out, raw.p, 192 * n, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  API_END
}

int sonic_srs_set_g2_points(sonic_srs_t* srs, const uint8_t* basis0, const uint8_t* basis1) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || !basis0 || !basis1) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  const long n = 2 * srs->d + 1;
  DevBuf raw(192 * n), err(4), h0(sizeof(G2Affine) * n), h1(sizeof(G2Affine) * n);
  HIP_OK(hipMemsetAsync(err.p, 0, 4, st));
  for (int b = 0; b < 2; b++) {
    HIP_OK(hipMemcpyAsync(raw.p, b ? basis1 : basis0, 192 * n, hipMemcpyHostToDevice, st));
    g2_points_from_bytes_enqueue(st, raw.as<uint8_t>(), (b ? h1 : h0).as<G2Affine>(), n, err.as<int>());
  }
  int herr = 0;
  HIP_OK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (herr) {
    set_error("sonic_srs_set_g2_points: %s", (herr & 1) ? "non-canonical coordinate" : (herr & 2) ? "point not on the twist" : (herr & 4) ? "point outside the order-r subgroup" : "point at infinity (h^{x^e}, h^{alpha x^e} are never the identity)");
    return SONIC_ERR_BAD_ENCODING;
  }
  std::lock_guard<std::mutex> g2(srs->g2_mu);
  srs->h = std::move(h0);
  srs->ha = std::move(h1);
  srs->wipe_trapdoor();
  API_END
}

int sonic_srs_has_g2(const sonic_srs_t* srs) {
  if (!srs) return 0;
  std::lock_guard<std::mutex> g2(srs->g2_mu);
  return (srs->h.p != nullptr || srs->have_trapdoor) ? 1 : 0;
}

// ---- on-disk SRS: "SONICSRS" | u32 version = 2 | u32 flags (bit 0: G2 half follows) | i64 d | basis0, basis1: (2d+1) x 96 B |
// [h basis0, h basis1: (2d+1) x 192 B], canonical affine encodings.  The reference has no persistence at all; this
// amortises SRS.new across runs, and with the G2 half a loaded SRS verifies as well as proves -- without the trapdoor.
int sonic_srs_save(const sonic_srs_t* srs, const char* path, int with_g2) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || !path || with_g2 < 0 || with_g2 > 2) return SONIC_ERR_INVALID_ARG;
  if (with_g2 == 2) with_g2 = sonic_srs_has_g2(srs) ? 1 : 0;          // "if the handle has (or can generate) it"
  if (with_g2) {
    std::lock_guard<std::mutex> g(call_mutex());
    int rc = srs_ensure_g2(srs, default_stream(), "sonic_srs_save");
    if (rc) return rc;
  }
  FILE* f = fopen(path, "wb");
  if (!f) { set_error("sonic_srs_save: cannot open %s", path); return SONIC_ERR_INVALID_ARG; }
  int64_t d = srs->d;
  bool ok = srs_file_write_header(f, d, with_g2 != 0);
  const int64_t n = 2 * d + 1, CH = 1 << 16;
  std::vector<uint8_t> buf(192 * (size_t)CH);
  for (int b = 0; b < (with_g2 ? 4 : 2) && ok; b++)
    for (int64_t i = 0; i < n && ok; i += CH) {
      int64_t m = n - i < CH ? n - i : CH;
      const size_t sz = b < 2 ? 96 : 192;
      int rc = b < 2 ? sonic_srs_get_points(srs, b, -d + i, m, buf.data()) : sonic_srs_get_g2_points(srs, b - 2, -d + i, m, buf.data());
      if (rc) { fclose(f); return rc; }
      ok = fwrite(buf.data(), sz, (size_t)m, f) == (size_t)m;
    }
  ok = (fclose(f) == 0) && ok;
  if (!ok) { set_error("sonic_srs_save: write to %s failed", path); return SONIC_ERR_INVALID_ARG; }
  API_END
}

int sonic_srs_load(const char* path, sonic_srs_t** out) { return sonic_srs_load_on(-1, path, out); }
int sonic_srs_load_on(int device, const char* path, sonic_srs_t** out) {
  API_BEGIN_ON(device)
  if (!path || !out) return SONIC_ERR_INVALID_ARG;
  SrsFile file;
  std::string why;
  const int frc = srs_file_read(path, file, why);             // header against the real file size before anything is allocated (srs_file.hpp)
  if (frc) { set_error("sonic_srs_load: %s", why.c_str()); return frc == 1 ? SONIC_ERR_INVALID_ARG : SONIC_ERR_BAD_ENCODING; }
  const int64_t d = file.d;
  const uint32_t flags = file.flags;
  std::vector<uint8_t>&b0 = file.g0, &b1 = file.g1, &h0 = file.h0, &h1 = file.h1;
  sonic_srs_t* s = nullptr;
  int rc = sonic_srs_from_points(d, b0.data(), b1.data(), &s);     // validates every point, rebuilds the window tables
  if (rc) return rc;
  if (flags & 1u) {
    rc = sonic_srs_set_g2_points(s, h0.data(), h1.data());
    if (rc) { sonic_srs_free(s); return rc; }
  }
  *out = s;
  API_END
}

// A replica of an SRS on another GPU of this process, copied device to device: both G1 bases with their window tables (so that the
// replica plans every MSM exactly like the original: sonic_prove_shared's ranks must agree on NB and W), the G2 half if present, the
// trapdoor if the G2 half has not been generated yet.  hipMemcpyPeer goes over xGMI where the GPUs are linked and through the host
// otherwise; nothing is validated again (the source was) and no table is rebuilt.
int sonic_srs_replicate(const sonic_srs_t* srs, int device, sonic_srs_t** out) {
  API_BEGIN_ON(device)
  if (!srs || !out) return SONIC_ERR_INVALID_ARG;
  DeviceCtx& ctx = current_ctx();
  std::lock_guard<std::mutex> g(ctx.call_mu);
  std::unique_ptr<sonic_srs> r(new sonic_srs());
  r->d = srs->d; r->device = ctx.dev;
  r->tab_c = srs->tab_c; r->tab_W = srs->tab_W; r->tab_endo = srs->tab_endo;
  r->g.alloc(srs->g.bytes); r->ga.alloc(srs->ga.bytes);
  HIP_OK(hipMemcpyPeer(r->g.p, ctx.dev, srs->g.p, srs->device, srs->g.bytes));
  HIP_OK(hipMemcpyPeer(r->ga.p, ctx.dev, srs->ga.p, srs->device, srs->ga.bytes));
  if (srs->ps.p) { r->ps.alloc(srs->ps.bytes); HIP_OK(hipMemcpyPeer(r->ps.p, ctx.dev, srs->ps.p, srs->device, srs->ps.bytes)); }
  if (srs->gs.p) { r->gs.alloc(srs->gs.bytes); HIP_OK(hipMemcpyPeer(r->gs.p, ctx.dev, srs->gs.p, srs->device, srs->gs.bytes)); }
  {
    std::lock_guard<std::mutex> g2(srs->g2_mu);
    if (srs->h.p) {
      r->h.alloc(srs->h.bytes); r->ha.alloc(srs->ha.bytes);
      HIP_OK(hipMemcpyPeer(r->h.p, ctx.dev, srs->h.p, srs->device, srs->h.bytes));
      HIP_OK(hipMemcpyPeer(r->ha.p, ctx.dev, srs->ha.p, srs->device, srs->ha.bytes));
    } else if (srs->have_trapdoor) {
      r->have_trapdoor = true; r->x_std = srs->x_std; r->alpha_std = srs->alpha_std;
    }
  }
  HIP_OK(hipDeviceSynchronize());
  *out = r.release();
  API_END
}

int sonic_abi_version(void) { return SONIC_ABI_VERSION; }
int sonic_msm_set_window(int c) { msm_set_window_override(c); return SONIC_OK; }
int sonic_srs_point_bytes(void) { return SONIC_SRS_POINT_BYTES; }

int sonic_msm_plan(const sonic_srs_t* srs, int64_t n, int* window_bits, int* windows, int* bucket_sets) {
  if (!srs || n < 0) return SONIC_ERR_INVALID_ARG;
  if (n >= MSM_MAX_TERMS) { set_error("MSM of %ld terms: at most 2^31 - 1 (entry encoding)", (long)n); return SONIC_ERR_INVALID_ARG; }
  MsmPlan pl = srs_msm_plan(srs, n);
  if (window_bits) *window_bits = pl.c;
  if (windows) *windows = pl.W;
  if (bucket_sets) *bucket_sets = pl.Wb;
  return SONIC_OK;
}

int sonic_msm_g1(const uint8_t* points, const uint8_t* scalars, int64_t n, uint8_t out_g1[96]) {
  API_BEGIN
  if (n < 0 || !out_g1 || (n > 0 && (!points || !scalars))) return SONIC_ERR_INVALID_ARG;
  CallLease lease;
  hipStream_t st = lease.st();
  const long m = n > 0 ? n : 1;
  DevBuf raw(96 * m), pts(sizeof(G1Affine) * m), sc(32 * m), err(4);
  HIP_OK(hipMemsetAsync(err.p, 0, 4, st));
  if (n > 0) {
    HIP_OK(hipMemcpyAsync(raw.p, points, 96 * n, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(sc.p, scalars, 32 * n, hipMemcpyHostToDevice, st));
    LAUNCH(k_points_from_bytes, ceil_div(n, 256), 256, 0, st, (const uint8_t*)raw.as<uint8_t>(), PointArrayMut{pts.as<char>(), (uint32_t)sizeof(G1Affine)}, (long)n,
           err.as<int>(), -1L);
    fr_check_enqueue(st, sc.as<Fr>(), n, err.as<int>());
  }
  int herr = 0;
  HIP_OK(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
  if (herr) { set_error("sonic_msm_g1: non-canonical input or point not on curve"); return SONIC_ERR_BAD_ENCODING; }
  msm_blocking(st, lease.ws(), msm_plan(n > 0 ? n : 1, /*fold=*/false), PointArray::packed(pts.as<G1Affine>()), sc.as<Fr>(), n, false, out_g1, nullptr);
  API_END
}

// ---- the same MSM in two halves, on a lane of its own (stream + bucket workspace + pinned result slot): submit queues scalar
// check, sort, accumulation and reduction and returns; collect waits, finishes on the host.  Two lanes used in turn keep the
// chip busy across consecutive MSMs: the sort and the (latency-bound) reduction of one run under the accumulation of the other.
struct sonic_msm_lane {
  int device = 0;                // the GPU the lane's stream and workspace live on; it serves SRS handles of that device
  hipStream_t st = nullptr;
  MsmWorkspace ws;
  DevBuf slot, err;
  MsmSlot* h_slot = nullptr;
  int* h_err = nullptr;
  int Wb = 0;
  bool in_flight = false;
  bool own_stream = true;        // false: the lane runs on a stream the caller owns (sonic_msm_lane_new_on_stream)
  // sonic_msm_g1_srs_multi keeps a pooled lane's exchange buffers across calls: this rank's scalar slice (host-scalar form), its full
  // bucket set, the slices it pulled from its peers, its device-side result
  DevBuf x_scalars, x_buckets, x_slices, x_part;
  std::mutex mu;
  ~sonic_msm_lane() {
    if (st) { (void)hipStreamSynchronize(st); if (own_stream) (void)hipStreamDestroy(st); }
    if (h_slot) (void)hipHostFree(h_slot);
    if (h_err) (void)hipHostFree(h_err);
  }
};

int sonic_msm_lane_new(sonic_msm_lane_t** out) { return sonic_msm_lane_new_on(-1, out); }
int sonic_msm_lane_new_on(int device, sonic_msm_lane_t** out) {
  API_BEGIN_ON(device)
  if (!out) return SONIC_ERR_INVALID_ARG;
  std::unique_ptr<sonic_msm_lane> l(new sonic_msm_lane());
  l->device = current_ctx().dev;
  HIP_OK(hipStreamCreateWithFlags(&l->st, hipStreamNonBlocking));
  l->slot.alloc(sizeof(MsmSlot));
  l->err.alloc(4);
  HIP_OK(hipHostMalloc((void**)&l->h_slot, sizeof(MsmSlot), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&l->h_err, 4, hipHostMallocDefault));
  *out = l.release();
  API_END
}
// A lane on a stream the caller owns -- e.g. torch.cuda.current_stream().cuda_stream, so that the lane's kernels, the caller's
// RCCL collectives on the bucket / partial buffers and the caller's copies are ordered by the stream and need no host
// synchronisation in between.  The stream must outlive the lane.
int sonic_msm_lane_new_on_stream(void* hip_stream, sonic_msm_lane_t** out) {
  // the lane lives on the stream's device (the null stream belongs to whatever device is current: the default device then)
  int sdev = -1;
  if (hip_stream && hipStreamGetDevice(static_cast<hipStream_t>(hip_stream), &sdev) != hipSuccess) { (void)hipGetLastError(); sdev = -1; }
  API_BEGIN_ON(sdev)
  if (!out) return SONIC_ERR_INVALID_ARG;
  std::unique_ptr<sonic_msm_lane> l(new sonic_msm_lane());
  l->device = current_ctx().dev;
  l->st = static_cast<hipStream_t>(hip_stream);
  l->own_stream = false;
  l->slot.alloc(sizeof(MsmSlot));
  l->err.alloc(4);
  HIP_OK(hipHostMalloc((void**)&l->h_slot, sizeof(MsmSlot), hipHostMallocDefault));
  HIP_OK(hipHostMalloc((void**)&l->h_err, 4, hipHostMallocDefault));
  *out = l.release();
  API_END
}
void sonic_msm_lane_free(sonic_msm_lane_t* l) {
  if (!l) return;
  try { DeviceScope scope(l->device); delete l; } catch (const HipFail&) { delete l; }
}
// a lane and an SRS handle meet in one call: both must live on the same GPU
static bool same_device(const sonic_msm_lane_t* l, const sonic_srs_t* srs, const char* who) {
  if (l->device == srs->device) return true;
  set_error("%s: the lane lives on device %d, the SRS on device %d", who, l->device, srs->device);
  return false;
}

static int msm_submit_common(sonic_msm_lane_t* l, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n, void* d_partial_out);
int sonic_msm_submit(sonic_msm_lane_t* l, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n) {
  return msm_submit_common(l, srs, basis, e0, d_scalars, n, nullptr);
}
// the same with the un-normalised 192-byte sum ALSO left in device memory (d_partial_out), queued on the lane's stream: the
// operand of a cross-rank all-gather that never visits the host.  Only plans that leave one window sum (window tables) can do
// that -- others need the host's Horner fold -- SONIC_ERR_INVALID_ARG otherwise (use sonic_msm_collect's out_partial then).
int sonic_msm_submit_dev_v2(sonic_msm_lane_t* l, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n, void* d_partial_out,
                            size_t out_bytes) {
  if (!d_partial_out) return SONIC_ERR_INVALID_ARG;
  if (out_bytes < sizeof(MsmSlot)) { set_error("sonic_msm_submit_dev_v2: the result takes %zu bytes of device memory (SONIC_G1_DEV_PARTIAL_BYTES), the buffer has %zu", sizeof(MsmSlot), out_bytes); return SONIC_ERR_INVALID_ARG; }
  return msm_submit_common(l, srs, basis, e0, d_scalars, n, d_partial_out);
}
// Retired symbols keep their OLD prototypes and refuse: up to round 3 these wrote 192 bytes to d_partial_out, since round 4 the result
// is SONIC_G1_DEV_PARTIAL_BYTES (12304) -- a caller built against the old header must get an error, not 12 KB written over its 192-byte
// buffer.  The _v2 forms take the buffer's size.
int sonic_msm_submit_dev(sonic_msm_lane_t*, const sonic_srs_t*, int, int64_t, const void*, int64_t, void*) {
  set_error("sonic_msm_submit_dev is retired (its result grew from 192 to %d bytes): use sonic_msm_submit_dev_v2, which takes the buffer size", SONIC_G1_DEV_PARTIAL_BYTES);
  return SONIC_ERR_INVALID_ARG;
}
static int msm_submit_common(sonic_msm_lane_t* l, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n, void* d_partial_out) {
  API_BEGIN_ON(l ? l->device : -1)
  if (!l || !srs || n < 0 || (n > 0 && !d_scalars) || (basis != 0 && basis != 1)) return SONIC_ERR_INVALID_ARG;
  if (!same_device(l, srs, "sonic_msm_submit")) return SONIC_ERR_INVALID_ARG;
  if (n > 0 && (e0 < -srs->d || e0 + n - 1 > srs->d)) { set_error("msm over SRS: exponent range [%ld, %ld] outside [-%ld, %ld]", (long)e0, (long)(e0 + n - 1), (long)srs->d, (long)srs->d); return SONIC_ERR_SRS_INDEX; }
  std::lock_guard<std::mutex> g(l->mu);
  if (l->in_flight) { set_error("sonic_msm_submit: the lane's previous MSM has not been collected"); return SONIC_ERR_INVALID_ARG; }
  hipStream_t st = l->st;
  const Fr* dsc = static_cast<const Fr*>(d_scalars);
  HIP_OK(hipMemsetAsync(l->err.p, 0, 4, st));
  fr_check_enqueue(st, dsc, n, l->err.as<int>());
  MsmPlan pl = srs_msm_plan(srs, n);
  pl.accum_block = 64;           // a lane's MSM runs alone or beside other lanes' MSMs, not inside a proof (msm.hpp)
  msm_enqueue(st, l->ws, pl, srs->basis(basis) + (e0 + srs->d), dsc, n, false, l->slot.as<MsmSlot>());
  l->Wb = pl.Wb;
  if (d_partial_out) HIP_OK(hipMemcpyAsync(d_partial_out, l->slot.p, sizeof(MsmSlot), hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(l->h_slot, l->slot.p, sizeof(MsmSlot), hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(l->h_err, l->err.p, 4, hipMemcpyDeviceToHost, st));
  l->in_flight = true;
  API_END
}

int sonic_msm_collect(sonic_msm_lane_t* l, uint8_t* out_g1, uint8_t* out_partial) {
  API_BEGIN_ON(l ? l->device : -1)
  if (!l) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(l->mu);
  if (!l->in_flight) { set_error("sonic_msm_collect: nothing was submitted"); return SONIC_ERR_INVALID_ARG; }
  l->in_flight = false;
  HIP_OK(hipStreamSynchronize(l->st));
  if (*l->h_err) { set_error("msm over SRS: non-canonical scalar"); return SONIC_ERR_BAD_ENCODING; }
  G1XYZZ sum = msm_finish_host(*l->h_slot);
  if (out_g1) g1_canonical_bytes_host(sum, out_g1);
  if (out_partial) memcpy(out_partial, &sum, sizeof sum);
  API_END
}

// a lane of the current device's pool (made on first need; workspaces and buffers stay with it), handed back when the holder dies
struct PooledLane {
  DeviceCtx& ctx;
  sonic_msm_lane_t* lane = nullptr;
  explicit PooledLane(DeviceCtx& c) : ctx(c) {
    {
      std::lock_guard<std::mutex> g(ctx.pool_mu);
      if (!ctx.lanes.empty()) { lane = static_cast<sonic_msm_lane_t*>(ctx.lanes.back()); ctx.lanes.pop_back(); }
    }
    if (!lane) {
      int rc = sonic_msm_lane_new_on(ctx.dev, &lane);
      if (rc) throw HipFail{rc};
    }
  }
  ~PooledLane() { if (lane) { std::lock_guard<std::mutex> g(ctx.pool_mu); ctx.lanes.push_back(lane); } }
  PooledLane(const PooledLane&) = delete;
  PooledLane& operator=(const PooledLane&) = delete;
};

// the blocking entry points run on one shared lane (pinned result slot, one host synchronisation per call)
static int msm_srs_common(const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, const uint8_t* h_scalars,
                          int64_t n, uint8_t* out96, uint8_t* out192) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || n < 0 || (basis != 0 && basis != 1)) return SONIC_ERR_INVALID_ARG;
  // a lane per call from the device's pool (the blocking MSMs of different host threads run side by side)
  PooledLane pooled(current_ctx());
  sonic_msm_lane_t* lane = pooled.lane;
  DevBuf sc;
  const void* dsc = d_scalars;
  if (h_scalars && n > 0) {
    sc.alloc(32 * n);
    HIP_OK(hipMemcpyAsync(sc.p, h_scalars, 32 * n, hipMemcpyHostToDevice, lane->st));
    dsc = sc.p;
  }
  if (n > 0 && !dsc) return SONIC_ERR_INVALID_ARG;
  int rc = sonic_msm_submit(lane, srs, basis, e0, dsc, n);
  if (rc) { (void)hipStreamSynchronize(lane->st); return rc; }       // (the upload of `sc` may still be in flight)
  return sonic_msm_collect(lane, out96, out192);       // waits for the stream: `sc` may go out of scope afterwards
  API_END
}

int sonic_msm_g1_srs(const sonic_srs_t* srs, int basis, int64_t e0, const uint8_t* scalars, int64_t n, uint8_t out_g1[96]) {
  return msm_srs_common(srs, basis, e0, nullptr, scalars, n, out_g1, nullptr);
}
int sonic_msm_g1_srs_dev(const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n, uint8_t out_g1[96]) {
  return msm_srs_common(srs, basis, e0, d_scalars, nullptr, n, out_g1, nullptr);
}
int sonic_msm_g1_srs_partial_dev(const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n, uint8_t out_partial[192]) {
  return msm_srs_common(srs, basis, e0, d_scalars, nullptr, n, nullptr, out_partial);
}

// ---- one MSM over several GPUs, sharded by BUCKET range (strong scaling) ----------------------------------------------------
// Every rank sorts and accumulates its term range into a full bucket set; the ranks exchange bucket ranges (all-to-all of
// slice_len x 192 B per pair); each rank adds the slices it received and reduces its range with the range's weights; the
// 192-byte partials are gathered and added as in the term-range scheme.  The reduction of the shared buckets -- which a
// term-range shard repeats in full on every rank -- is divided by the number of ranks this way.
int sonic_msm_exchange_layout(const sonic_srs_t* srs, int world, int64_t* n_buckets, int64_t* slice_len) {
  if (!srs || world < 1 || !n_buckets || !slice_len) return SONIC_ERR_INVALID_ARG;
  if (srs->tab_W <= 1 || srs->tab_endo) { set_error("bucket exchange needs the full window tables of the SRS (one shared bucket set per MSM)"); return SONIC_ERR_INVALID_ARG; }
  const int64_t NB = 1LL << (srs->tab_c - 1);
  int64_t S = (NB + world - 1) / world;
  S = (S + MSM_SLICE_QUANTUM - 1) / MSM_SLICE_QUANTUM * MSM_SLICE_QUANTUM;
  *n_buckets = NB;
  *slice_len = S;
  return SONIC_OK;
}

int sonic_msm_accumulate_dev(sonic_msm_lane_t* l, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n,
                             void* d_buckets, int64_t capacity) {
  API_BEGIN_ON(l ? l->device : -1)
  if (!l || !srs || n < 0 || (n > 0 && !d_scalars) || (basis != 0 && basis != 1) || !d_buckets) return SONIC_ERR_INVALID_ARG;
  if (!same_device(l, srs, "sonic_msm_accumulate_dev")) return SONIC_ERR_INVALID_ARG;
  if (n > 0 && (e0 < -srs->d || e0 + n - 1 > srs->d)) { set_error("msm over SRS: exponent range [%ld, %ld] outside [-%ld, %ld]", (long)e0, (long)(e0 + n - 1), (long)srs->d, (long)srs->d); return SONIC_ERR_SRS_INDEX; }
  if (srs->tab_W <= 1 || srs->tab_endo) { set_error("sonic_msm_accumulate_dev needs the full window tables of the SRS"); return SONIC_ERR_INVALID_ARG; }
  const int64_t NB = 1LL << (srs->tab_c - 1);
  if (capacity < NB) { set_error("sonic_msm_accumulate_dev: room for %ld buckets, the plan has %ld", (long)capacity, (long)NB); return SONIC_ERR_INVALID_ARG; }
  std::lock_guard<std::mutex> g(l->mu);
  hipStream_t st = l->st;
  const Fr* dsc = static_cast<const Fr*>(d_scalars);
  HIP_OK(hipMemsetAsync(l->err.p, 0, 4, st));
  fr_check_enqueue(st, dsc, n, l->err.as<int>());
  // always over the tables (a term-range share may be small against the bucket set; every rank must fill the SAME buckets)
  MsmPlan pl = msm_plan_tables(n > 0 ? n : 1, srs->tab_c, srs->tab_W, 2 * srs->d + 1);
  pl.accum_block = 64;
  MsmJob job{srs->basis(basis) + (e0 + srs->d), dsc, (long)n, nullptr};
  msm_enqueue_batch(st, l->ws, pl, &job, 1, false, static_cast<G1XYZZ*>(d_buckets));
  if (capacity > NB) HIP_OK(hipMemsetAsync(static_cast<G1XYZZ*>(d_buckets) + NB, 0, sizeof(G1XYZZ) * (size_t)(capacity - NB), st));   // padding = infinity
  HIP_OK(hipMemcpyAsync(l->h_err, l->err.p, 4, hipMemcpyDeviceToHost, st));
  API_END
}

int sonic_msm_reduce_slices_dev(sonic_msm_lane_t*, const sonic_srs_t*, const void*, int, int64_t, int64_t, void*) {
  set_error("sonic_msm_reduce_slices_dev is retired (its result grew from 192 to %d bytes): use sonic_msm_reduce_slices_dev_v2, which takes the buffer size", SONIC_G1_DEV_PARTIAL_BYTES);
  return SONIC_ERR_INVALID_ARG;
}
int sonic_msm_reduce_slices_dev_v2(sonic_msm_lane_t* l, const sonic_srs_t* srs, const void* d_slices, int k, int64_t slice_len, int64_t bucket_base,
                                   void* d_partial_out, size_t out_bytes) {
  API_BEGIN_ON(l ? l->device : -1)
  if (!l || !srs || !d_slices || k < 1 || !d_partial_out || slice_len < 1 || bucket_base < 0) return SONIC_ERR_INVALID_ARG;
  if (out_bytes < sizeof(MsmSlot)) { set_error("sonic_msm_reduce_slices_dev_v2: the result takes %zu bytes of device memory (SONIC_G1_DEV_PARTIAL_BYTES), the buffer has %zu", sizeof(MsmSlot), out_bytes); return SONIC_ERR_INVALID_ARG; }
  if (!same_device(l, srs, "sonic_msm_reduce_slices_dev_v2")) return SONIC_ERR_INVALID_ARG;
  if (slice_len % MSM_SLICE_QUANTUM || bucket_base % MSM_SLICE_QUANTUM) { set_error("sonic_msm_reduce_slices_dev: slice length and base must be multiples of %ld (sonic_msm_exchange_layout)", (long)MSM_SLICE_QUANTUM); return SONIC_ERR_INVALID_ARG; }
  std::lock_guard<std::mutex> g(l->mu);
  hipStream_t st = l->st;
  msm_reduce_slices_enqueue(st, l->ws, static_cast<const G1XYZZ*>(d_slices), k, slice_len, bucket_base, srs->tab_c, l->slot.as<MsmSlot>());
  HIP_OK(hipMemcpyAsync(d_partial_out, l->slot.p, sizeof(MsmSlot), hipMemcpyDeviceToDevice, st));
  API_END
}

// waits for what the lane has queued (accumulate / reduce / submit_dev) and reports a non-canonical scalar
int sonic_msm_lane_sync(sonic_msm_lane_t* l) {
  API_BEGIN_ON(l ? l->device : -1)
  if (!l) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(l->mu);
  HIP_OK(hipStreamSynchronize(l->st));
  if (*l->h_err) { *l->h_err = 0; set_error("msm over SRS: non-canonical scalar"); return SONIC_ERR_BAD_ENCODING; }
  API_END
}

// ---- ONE MSM over several GPUs of this process (include/sonic_hip.h, "N GPUs from ONE host process") -------------------------------
// One host thread per replica; thread r owns a lane on srs[r]'s GPU.  mode 0: every GPU a whole MSM over its term range, the host adds
// the partial sums.  mode 1: every GPU accumulates its term range into a full bucket set, waits for the others at a host barrier, pulls
// ITS bucket range from every peer (hipMemcpyPeerAsync on its own lane's stream: xGMI is point to point, the world - 1 pulls of a GPU
// run on world - 1 links), adds the slices, reduces 1/world of the buckets and reports its device-side result; the host adds those.
namespace {
struct HostBarrier {
  std::mutex mu; std::condition_variable cv; int n, waiting = 0, phase = 0;
  explicit HostBarrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> l(mu);
    const int ph = phase;
    if (++waiting == n) { waiting = 0; phase++; cv.notify_all(); return; }
    cv.wait(l, [&] { return phase != ph; });
  }
};
}
static int msm_multi(const sonic_srs_t* const* srs, int world, int basis, const int64_t* e0, const void* const* d_scalars, const uint8_t* const* h_scalars,
                     const int64_t* n, int mode, uint8_t out_g1[96]) {
  if (!srs || world < 1 || world > 64 || !out_g1 || (basis != 0 && basis != 1) || (mode != 0 && mode != 1)) return SONIC_ERR_INVALID_ARG;
  for (int r = 0; r < world; r++) {
    if (!srs[r] || n[r] < 0 || (n[r] > 0 && !(d_scalars ? d_scalars[r] : (const void*)h_scalars[r]))) return SONIC_ERR_INVALID_ARG;
    if (srs[r]->d != srs[0]->d) { set_error("sonic_msm_g1_srs_multi: replica %d has d = %ld, replica 0 d = %ld", r, (long)srs[r]->d, (long)srs[0]->d); return SONIC_ERR_INVALID_ARG; }
  }
  int64_t NB = 0, S = 0;
  if (mode == 1 && world > 1) {
    for (int r = 0; r < world; r++) {
      int64_t nb_r = 0, s_r = 0;
      int rc = sonic_msm_exchange_layout(srs[r], world, &nb_r, &s_r);
      if (rc) return rc;
      if (r == 0) { NB = nb_r; S = s_r; }
      else if (nb_r != NB || s_r != S) { set_error("sonic_msm_g1_srs_multi: replica %d runs another MSM plan than replica 0 (window tables differ)", r); return SONIC_ERR_INVALID_ARG; }
    }
  }
  const bool exchange = mode == 1 && world > 1;
  std::vector<int> rcs((size_t)world, SONIC_OK);
  std::vector<std::string> errs((size_t)world);
  std::vector<G1XYZZ> partial((size_t)world, G1XYZZ::inf());
  std::vector<uint8_t> blobs(exchange ? sizeof(MsmSlot) * (size_t)world : 0);
  std::vector<void*> bucket_ptr((size_t)world, nullptr);       // each rank's full bucket set (mode 1), read by its peers
  HostBarrier bar(world);
  auto body = [&](int r) {
    int rc = SONIC_OK;
    const int dev = srs[r]->device;
    bool at_barrier_1 = false, at_barrier_2 = false;
    auto fail = [&](int code) { rc = code; char b[512]; sonic_last_error(b, sizeof b); errs[(size_t)r] = b; };
    try {
      DeviceScope scope(dev);
      PooledLane pooled(current_ctx());          // the device's pooled lanes keep their workspaces and exchange buffers from call to call
      sonic_msm_lane_t* lane = pooled.lane;
      do {
        const void* sc = d_scalars ? d_scalars[r] : nullptr;
        if (!d_scalars && n[r] > 0) {
          lane->x_scalars.ensure(32 * (size_t)n[r]);
          HIP_OK(hipMemcpyAsync(lane->x_scalars.p, h_scalars[r], 32 * (size_t)n[r], hipMemcpyHostToDevice, lane->st));
          sc = lane->x_scalars.p;
        }
        if (!exchange) {
          uint8_t part[192];
          if ((rc = sonic_msm_submit(lane, srs[r], basis, e0[r], sc, n[r])) || (rc = sonic_msm_collect(lane, nullptr, part))) { fail(rc); (void)hipStreamSynchronize(lane->st); break; }
          memcpy(&partial[(size_t)r], part, 192);
          break;
        }
        const size_t cap = (size_t)world * (size_t)S;
        lane->x_buckets.ensure(cap * sizeof(G1XYZZ)); lane->x_slices.ensure(cap * sizeof(G1XYZZ)); lane->x_part.ensure(sizeof(MsmSlot));
        if ((rc = sonic_msm_accumulate_dev(lane, srs[r], basis, e0[r], sc, n[r], lane->x_buckets.p, (int64_t)cap)) || (rc = sonic_msm_lane_sync(lane))) fail(rc);
        if (rc) (void)hipStreamSynchronize(lane->st);
        bucket_ptr[(size_t)r] = rc ? nullptr : lane->x_buckets.p;
        bar.wait(); at_barrier_1 = true;                           // every rank's buckets are complete (or it has failed)
        bool all = true;
        for (int q = 0; q < world; q++) all = all && bucket_ptr[(size_t)q] != nullptr;
        if (all) {
          try {
            // direct reads over xGMI where the GPUs are linked: ask once per pair (a runtime that refuses, or has it on already, still
            // copies -- staged -- so the answer is not an error here)
            for (int q = 0; q < world; q++)
              if (srs[q]->device != dev) { if (hipDeviceEnablePeerAccess(srs[q]->device, 0) != hipSuccess) (void)hipGetLastError(); }
            for (int q = 0; q < world; q++)                        // slice r of rank q -> slot q of my [world][S] buffer
              HIP_OK(hipMemcpyPeerAsync(lane->x_slices.as<G1XYZZ>() + (size_t)q * S, dev, static_cast<const G1XYZZ*>(bucket_ptr[(size_t)q]) + (size_t)r * S,
                                        srs[q]->device, (size_t)S * sizeof(G1XYZZ), lane->st));
          } catch (const HipFail& f) { fail(f.code); }
          if (!rc && ((rc = sonic_msm_reduce_slices_dev_v2(lane, srs[r], lane->x_slices.p, world, S, (int64_t)r * S, lane->x_part.p, sizeof(MsmSlot))) ||
                      (rc = sonic_msm_lane_sync(lane)))) fail(rc);
          if (!rc) {
            try { HIP_OK(hipMemcpy(&blobs[sizeof(MsmSlot) * (size_t)r], lane->x_part.p, sizeof(MsmSlot), hipMemcpyDeviceToHost)); } catch (const HipFail& f) { fail(f.code); }
          }
        } else if (!rc) { rc = -1; errs[(size_t)r] = "a peer rank failed before the bucket exchange"; }      // (secondary: the peer's own status is the call's)
        (void)hipStreamSynchronize(lane->st);
        bar.wait(); at_barrier_2 = true;                           // nobody re-uses its buckets while a peer still reads them
      } while (false);
    } catch (const HipFail& f) { fail(f.code); }
    catch (const std::exception& e) { rc = SONIC_ERR_HIP; errs[(size_t)r] = e.what(); }       // (nothing may leave a thread's body: std::terminate)
    catch (...) { rc = SONIC_ERR_HIP; errs[(size_t)r] = "unknown exception"; }
    if (exchange) { if (!at_barrier_1) bar.wait(); if (!at_barrier_2) bar.wait(); }
    rcs[(size_t)r] = rc;
  };
  if (world == 1) body(0);
  else {
    ThreadGroup th;
    for (int r = 0; r < world; r++) th.emplace_back(body, r);
    for (auto& t : th) t.join();
  }
  for (int pass = 0; pass < 2; pass++)               // a rank's own failure first; "a peer failed" only if nothing else explains it
    for (int r = 0; r < world; r++)
      if (pass == 0 ? rcs[(size_t)r] > 0 : rcs[(size_t)r] != 0) {
        set_error("sonic_msm_g1_srs_multi, rank %d (device %d): %s", r, srs[r]->device, errs[(size_t)r].c_str());
        return rcs[(size_t)r] > 0 ? rcs[(size_t)r] : SONIC_ERR_HIP;
      }
  if (exchange) return sonic_g1_sum_dev_partials(blobs.data(), world, out_g1);
  G1XYZZ acc = G1XYZZ::inf();
  for (int r = 0; r < world; r++) acc = g1_add(acc, partial[(size_t)r]);
  g1_canonical_bytes_host(acc, out_g1);
  return SONIC_OK;
}

int sonic_msm_g1_srs_multi_dev(const sonic_srs_t* const* srs, int world, int basis, const int64_t* e0, const void* const* d_scalars, const int64_t* n, int mode,
                               uint8_t out_g1[96]) {
  if (!e0 || !d_scalars || !n) return SONIC_ERR_INVALID_ARG;
  try { return msm_multi(srs, world, basis, e0, d_scalars, nullptr, n, mode, out_g1); }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}
int sonic_msm_g1_srs_multi(const sonic_srs_t* const* srs, int world, int basis, int64_t e0, const uint8_t* scalars, int64_t n, int mode, uint8_t out_g1[96]) {
  if (world < 1 || world > 64 || n < 0 || (n > 0 && !scalars)) return SONIC_ERR_INVALID_ARG;
  std::vector<int64_t> e((size_t)world), cnt((size_t)world);
  std::vector<const uint8_t*> hs((size_t)world);
  for (int r = 0; r < world; r++) {                              // contiguous term ranges, as even as possible
    const int64_t lo = n * r / world, hi = n * (r + 1) / world;
    e[(size_t)r] = e0 + lo; cnt[(size_t)r] = hi - lo; hs[(size_t)r] = scalars + 32 * (size_t)lo;
  }
  try { return msm_multi(srs, world, basis, e.data(), nullptr, hs.data(), cnt.data(), mode, out_g1); }
  catch (const std::exception& e2) { set_error("%s", e2.what()); return SONIC_ERR_HIP; }
}

// the device-side partials of sonic_msm_submit_dev / sonic_msm_reduce_slices_dev (SONIC_G1_DEV_PARTIAL_BYTES each: what the bulk kernels
// leave of an MSM, a handful of points the host folds) -> their sum, normalised
static_assert(sizeof(MsmSlot) == SONIC_G1_DEV_PARTIAL_BYTES, "SONIC_G1_DEV_PARTIAL_BYTES is the size of a device-side MSM result");
int sonic_g1_sum_dev_partials(const uint8_t* blobs, int k, uint8_t out_g1[96]) {
  if (!blobs || k < 1 || !out_g1) return SONIC_ERR_INVALID_ARG;
  G1XYZZ acc = G1XYZZ::inf();
  std::unique_ptr<MsmSlot> s(new MsmSlot());
  for (int i = 0; i < k; i++) {
    memcpy(s.get(), blobs + sizeof(MsmSlot) * (size_t)i, sizeof(MsmSlot));
    if (s->W < 0 || s->W > MSM_MAX_WINDOWS || (s->pad1 == 1 && s->W >= MSM_MAX_WINDOWS) || (s->pad1 == 2 && s->W >= MSM_MAX_WINDOWS - MSM_ENDO_SLOT_OFFSET) || s->pad1 < 0 || s->pad1 > 2 || s->c < 0 || s->c > 32) { set_error("sonic_g1_sum_dev_partials: blob %d is not a device-side MSM result", i); return SONIC_ERR_INVALID_ARG; }
    acc = g1_add(acc, msm_finish_host(*s));
  }
  g1_canonical_bytes_host(acc, out_g1);
  return SONIC_OK;
}

int sonic_g1_sum_partials(const uint8_t* partials, int k, uint8_t out_g1[96]) {
  if (!partials || k < 1 || !out_g1) return SONIC_ERR_INVALID_ARG;
  G1XYZZ acc = G1XYZZ::inf();
  for (int i = 0; i < k; i++) { G1XYZZ p; memcpy(&p, partials + sizeof(G1XYZZ) * i, sizeof p); acc = g1_add(acc, p); }
  g1_canonical_bytes_host(acc, out_g1);
  return SONIC_OK;
}

// device memory for callers without a HIP binding: _on allocates on a named GPU; free / upload / download find the pointer's device
static int device_of_pointer(const void* p) {
  hipPointerAttribute_t a;
  if (p && hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
  (void)hipGetLastError();
  return -1;
}
int sonic_dev_alloc(size_t bytes, void** out) { return sonic_dev_alloc_on(-1, bytes, out); }
int sonic_dev_alloc_on(int device, size_t bytes, void** out) { API_BEGIN_ON(device) if (!out) return SONIC_ERR_INVALID_ARG; HIP_OK(hipMalloc(out, bytes ? bytes : 16)); API_END }
int sonic_dev_free(void* p) { API_BEGIN_ON(device_of_pointer(p)) HIP_OK(hipFree(p)); API_END }
int sonic_dev_upload(void* dst, const void* src, size_t bytes) { API_BEGIN_ON(device_of_pointer(dst)) HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); API_END }
int sonic_dev_download(void* dst, const void* src, size_t bytes) { API_BEGIN_ON(device_of_pointer(src)) HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); API_END }

int sonic_profile_enable(int on) { profiler().on = on != 0; return SONIC_OK; }
int sonic_profile_reset(void) { profiler().reset(); return SONIC_OK; }
int sonic_profile_get(const char* kernel, double* total_ms, int64_t* launches) {
  if (!kernel) return SONIC_ERR_INVALID_ARG;
  profiler().collect();
  std::lock_guard<std::mutex> g(profiler().mu);
  auto it = profiler().totals.find(kernel);
  if (total_ms) *total_ms = it == profiler().totals.end() ? 0.0 : it->second.first;
  if (launches) *launches = it == profiler().totals.end() ? 0 : it->second.second;
  return SONIC_OK;
}
int sonic_profile_names(char* buf, size_t cap) {
  if (!buf || cap == 0) return SONIC_ERR_INVALID_ARG;
  profiler().collect();
  std::lock_guard<std::mutex> g(profiler().mu);
  std::string s;
  for (auto& kv : profiler().totals) { s += kv.first; s += "\n"; }
  strncpy(buf, s.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return SONIC_OK;
}

}  // extern "C"
