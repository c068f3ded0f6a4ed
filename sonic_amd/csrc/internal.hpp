// Cross-translation-unit declarations inside libsonic_hip.so.
#pragma once
#include <atomic>
#include "common.hpp"
#include "g1.hpp"
#include "msm.hpp"

struct sonic_srs;
struct sonic_prover;

namespace sonic {

struct NttTables;

// ---- devices (api.hip) ------------------------------------------------------------------------------------------------------------
// One host process may drive every GPU of the node (round 5): each handle -- SRS, prover, MSM lane -- is bound to the device it was
// made on, and what used to be process-wide state (the default stream, the mutex of the state-changing calls, the leased call
// contexts, the pool of blocking-MSM lanes, the twiddle tables and scratch of the stand-alone NTT entry points, per-device kernel
// attributes) lives in a DeviceCtx per ordinal.  Every entry point opens a DeviceScope: it makes the handle's device the calling
// thread's HIP device (hipSetDevice is per thread) for the duration of the call and puts the previous one back afterwards.
struct CallCtx;
struct DeviceCtx {
  int dev = -1;
  hipStream_t stream = nullptr;          // the device's default stream for the serialised (state-changing) calls
  std::mutex call_mu;                    // SRS construction, the lazy G2 half, the shared NTT tables of the stand-alone entry points
  std::mutex pool_mu;
  std::vector<CallCtx*> pool;            // leased call contexts (CallLease)
  std::vector<void*> lanes;              // idle lanes of the blocking MSM entry points (sonic_msm_lane*, under pool_mu)
  NttTables* ntt = nullptr;              // sonic_ntt_fr / sonic_poly_mul_fr[_dev] (under call_mu)
  DevBuf mul_a, mul_b, mul_flags;        // scratch of sonic_poly_mul_fr_dev (under call_mu)
  std::atomic<int> sort_staged{-1};      // msm.hip: the LDS-staged sort passes got their dynamic-LDS attribute on this device (-1: not asked yet)
  std::atomic<int> ntt_big{-1};          // ntt.hip: the same for the 128-KB block of k_ntt_wide_big
  // twiddle tables of the prover handles, one set per transform size, shared by every handle on the device and never freed (64 + 64 MB
  // at 2^21 points: two streaming handles used to hold one copy each) (under pool_mu)
  std::map<int, NttTables*> prover_ntt;
  // the idle prover shells the one-shot sonic_prove keeps for later calls with the same SRS and circuit shape (prove.hip): at most
  // ONE_SHOT_SHELLS, the oldest goes when another is parked
  std::mutex one_shot_mu;
  std::vector<void*> one_shot;
};
const NttTables& device_ntt_tables(int log2n);          // of the current device; built on first use (blocks until they are complete)
void drop_one_shot_of(const sonic_srs* s);              // an SRS handle is going away: the parked one-shot shells over it go first
struct OneShotShell { const sonic_srs* srs; sonic_prover* p; };      // an entry of DeviceCtx::one_shot (prove.hip)
void unlink_one_shot_of(const sonic_srs* s);            // the same when no device scope can be opened: the shells are forgotten (leaked), never matched again
// dev < 0: the process's default device (sonic_init, else LOCAL_RANK % device count, else 0).  Throws HipFail{SONIC_ERR_NO_DEVICE}
// without a GPU -- the library has no CPU fallback -- and HipFail{SONIC_ERR_INVALID_ARG} for an ordinal the node does not have.
class DeviceScope {
 public:
  explicit DeviceScope(int dev = -1);
  ~DeviceScope();
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
  DeviceCtx& ctx() const { return *ctx_; }
 private:
  DeviceCtx* ctx_;
  DeviceCtx* prev_ctx_;
  int prev_dev_;
};
DeviceCtx& current_ctx();                 // the innermost DeviceScope of the calling thread
int default_device_ordinal();             // -1 before the first call that needed a device
inline hipStream_t default_stream() { return current_ctx().stream; }
// serialises the calls that CHANGE shared state on the current device; the read-only entry points over an SRS (commitPoly, openPoly,
// hscProve, the blocking MSMs, srs_get_points) run on a leased context instead
inline std::mutex& call_mutex() { return current_ctx().call_mu; }
// A stream + MSM workspace of its own for the duration of one call: read-only calls on a shared SRS are re-entrant.
// Contexts are pooled per device and grow with the number of host threads that are inside the library at once.
struct CallCtx { hipStream_t st = nullptr; MsmWorkspace ws; };
class CallLease {
 public:
  CallLease();
  ~CallLease();
  CallLease(const CallLease&) = delete;
  CallLease& operator=(const CallLease&) = delete;
  hipStream_t st() const { return c_->st; }
  MsmWorkspace& ws() const { return c_->ws; }
 private:
  CallCtx* c_;
  DeviceCtx* owner_;
};

// SRS handle internals (api.hip)
PointArray srs_basis(const sonic_srs* s, int b);            // table 0 of a basis, slot e + d; window table w follows at + w (2d+1)
PointArrayMut srs_basis_mut(sonic_srs* s, int b);
int64_t srs_d(const sonic_srs* s);
int srs_device(const sonic_srs* s);          // -1 for a null handle (= the default device: the null check then reports the argument)
sonic_srs* srs_alloc(int64_t d);
// the handle's Fiat-Shamir id (fs.hpp), made once by `make` and kept in the handle
int srs_cached_id(const sonic_srs* s, int (*make)(const sonic_srs*, uint8_t*), uint8_t out[32]);

// encodings (api.hip)
void fr_to_mont_enqueue(hipStream_t st, Fr* d, long n, int* d_err);
void fr_from_mont_enqueue(hipStream_t st, Fr* d, long n);
void fr_check_enqueue(hipStream_t st, const Fr* d, long n, int* d_err);
void msm_blocking(hipStream_t st, MsmWorkspace& ws, const MsmPlan& pl, PointArray d_pts, const Fr* d_sc, long n, bool mont,
                  uint8_t* out96, uint8_t* out_partial192);
int srs_tab_c(const sonic_srs* s);
int srs_tab_W(const sonic_srs* s);
bool srs_tab_endo(const sonic_srs* s);
MsmPlan srs_msm_plan(const sonic_srs* s, long n);
// fills window tables 1 .. W-1 of both bases from table 0 (srs.hip)
void srs_build_tables(hipStream_t st, sonic_srs* s);
PointArrayMut srs_prefix_mut(sonic_srs* s);      // running sums of the alpha basis (p == nullptr: not held)
PointArray srs_prefix(const sonic_srs* s);
PointArrayMut srs_sym_mut(sonic_srs* s);         // symmetric sums A[e] + A[-e] of the alpha basis, laid out like a basis with its window tables
PointArray srs_sym(const sonic_srs* s);
void srs_set_trapdoor(sonic_srs* s, const Fr& x_std, const Fr& alpha_std);
struct G2Affine;
// G2 half (srs_g2.hip)
void srs_generate_g2(hipStream_t st, long d, const Fr& x_std, const Fr& alpha_std, G2Affine* h0, G2Affine* h1);
void g2_points_to_bytes_enqueue(hipStream_t st, const G2Affine* in, uint8_t* d_out, long n);
void g2_points_from_bytes_enqueue(hipStream_t st, const uint8_t* d_in, G2Affine* out, long n, int* d_err);

// SRS generation (srs.hip): fills both bases of `s` from x, alpha (standard-form Fr on the host)
void srs_generate(hipStream_t st, sonic_srs* s, const Fr& x_std, const Fr& alpha_std);

// NTT (ntt.hip).  Data in Montgomery form, in place.  forward: natural -> bit-reversed;
// inverse: bit-reversed -> natural, scaled by 1/n.
struct NttTables {
  int log2n = 0;
  DevBuf fwd, inv;    // stage-major twiddles, 2^log2n entries each: [2^L - 2^(L-s) + j] = w^(+-j 2^s), j < 2^(L-1-s) (Montgomery; ntt.hip)
  DevBuf ninv;        // (2^k)^-1, k = 0..32
  void ensure(hipStream_t st, int log2n);
};
void ntt_forward_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n);
void ntt_inverse_enqueue(hipStream_t st, const NttTables& tw, Fr* d, int log2n);
void ntt_inverse_of_product_enqueue(hipStream_t st, const NttTables& tw, Fr* d, const Fr* other, int log2n);
void fr_pointwise_mul_enqueue(hipStream_t st, Fr* a, const Fr* b, long n);
void fr_scale_enqueue(hipStream_t st, Fr* a, long n, const Fr* d_s);
void fr_bitrev_permute_enqueue(hipStream_t st, Fr* d, int log2n);

// polynomial kernels (poly.hip); all Fr arrays Montgomery, dense over an exponent range
// out[i] = in[i] * x^(e0 + i)
void poly_scale_powers_enqueue(hipStream_t st, const Fr* in, Fr* out, long n, long e0, const Fr* d_x, const Fr* d_xinv);
// inclusive prefix sums in Fr, in place; tmp >= ceil(n/1024)+1 Fr
void poly_prefix_sum_enqueue(hipStream_t st, Fr* d, long n, DevBuf& tmp);
// quotient of (f - f(z)) / (X - z) from d = f_e z^e and its prefix sums P (see poly.hip)
void poly_quotient_enqueue(hipStream_t st, const Fr* prefix, Fr* q, long n, long lo, const Fr* d_z, const Fr* d_zinv);

}  // namespace sonic
