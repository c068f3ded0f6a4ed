/* sonic_hip.h -- C ABI of libsonic_hip.so, the MI355X-native Sonic prover hot path.
 *
 * The reference (sdiehl/sonic, Haskell) has no FFI: its public surface is the exposed modules of
 * sonic.cabal:31-37.  Each entry point below names the Haskell function it stands in for; a
 * `foreign import ccall` shim over these symbols re-creates Sonic.SRS /
 * Sonic.CommitmentScheme / Sonic.Protocol 1:1.
 *
 * Value encodings (standard form, never Montgomery):
 *   Fr  : 32 bytes little-endian integer < r
 *   G1  : 96 bytes  x || y, each 48 bytes little-endian < q; the point at infinity (`mempty`) is
 *         96 zero bytes ((0,0) is not on y^2 = x^3 + 4)
 *   sparse Laurent polynomial (poly's VLaurent as `GHC.Exts.toList` yields it,
 *         CommitmentScheme.hs:33,48): n_terms exponents (int64) + n_terms Fr coefficients;
 *         exponents need not be sorted, repeated exponents are summed, zero coefficients allowed
 *   gate weights wL, wR, wO (Bulletproofs GateWeights, lists of Q rows of n): dense Q x n
 *         row-major Fr
 *   transcript: the prover's `rnd` draws made explicit, in draw order (Protocol.hs:58,66,76,
 *         84-85; Signature.hs:48,60):  c_{n+1..n+4}, y, z, y_1..y_Q, z_1..z_Q, u, v  = 8 + 2Q Fr
 *   proof: record order of `Proof` (Protocol.hs:28-38) then `HscProof` (Signature.hs:22-29):
 *         R, T, a, Wa, b, Wb, Wt, s, [S_j, s_j, W_j]_j, [s'_j, W'_j, Q_j]_j, Qv, C, u, v
 *         = (7+4Q)*96 + (5+2Q)*32 bytes
 *
 * Errors: the reference panics (Protocol.hs:55, CommitmentScheme.hs:70-73) or hits `fromJust`
 * (CommitmentScheme.hs:44); here every function returns a status and never aborts the process.
 * sonic_last_error() gives the thread's last message.  There is no CPU fallback: without a HIP
 * device every call returns SONIC_ERR_NO_DEVICE.
 *
 * Threading: an SRS handle is immutable after construction and may be shared; a prover handle
 * owns its stream and workspace and serves one call at a time.
 *
 * Devices (round 5): one host process may drive every GPU of the node.  Every handle -- SRS, prover, MSM lane -- lives on the GPU it
 * was made on (the *_on constructors name it; the others use the default device: sonic_init, else LOCAL_RANK, else 0) and every
 * call on a handle runs there, whatever the calling thread's current HIP device is (which the call leaves as it found it).
 * sonic_prove_shared / sonic_prove_batch / sonic_msm_g1_srs_multi run ONE call over handles on several GPUs with one host thread
 * per device inside the library: no launcher, no torch, no RCCL.
 *
 * ABI version: sonic_abi_version().  An entry point whose argument list or buffer sizes change gets a NEW symbol (…_v2); the old
 * symbol keeps its old prototype and returns SONIC_ERR_INVALID_ARG, so that a caller built against an older header gets a status
 * instead of a memory overwrite.
 */
#ifndef SONIC_HIP_H
#define SONIC_HIP_H
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

enum {
  SONIC_OK = 0,
  SONIC_ERR_D_TOO_SMALL = 1,      /* Protocol.hs:54-55  "Parameter d is not large enough" */
  SONIC_ERR_SRS_INDEX = 2,        /* CommitmentScheme.hs:70-73 "... is not long enough", incl. the
                                     e' = 0 hole of the alpha basis (index -1) */
  SONIC_ERR_BAD_ENCODING = 3,     /* non-canonical Fr / Fq, point not on the curve */
  SONIC_ERR_INEXACT_DIVISION = 4, /* CommitmentScheme.hs:44 fromJust; also z = 0 with negative exponents */
  SONIC_ERR_HIP = 5,
  SONIC_ERR_NO_DEVICE = 6,
  SONIC_ERR_INVALID_ARG = 7
};

#define SONIC_FR_BYTES 32
#define SONIC_G1_BYTES 96
#define SONIC_G1_PARTIAL_BYTES 192   /* un-normalised XYZZ accumulator exchanged between ranks */
/* what the bulk kernels leave of an MSM in DEVICE memory: a header and up to 64 points that the host folds in microseconds (window sums
 * of a Horner walk, or the bit sums of the round-4 bucket reduction) -- the operand of the cross-rank all-gather that never visits the
 * host (sonic_msm_submit_dev, sonic_msm_reduce_slices_dev), summed by sonic_g1_sum_dev_partials */
#define SONIC_G1_DEV_PARTIAL_BYTES 12304

typedef struct sonic_srs sonic_srs_t;
typedef struct sonic_prover sonic_prover_t;

/* ---- library ---- */
#define SONIC_ABI_VERSION 6
/* the SONIC_ABI_VERSION the library was built as; no device needed.  History: 5 = round 5 (devices; sonic_msm_submit_dev_v2,
 * sonic_msm_reduce_slices_dev_v2 and sonic_fs_challenges_v2 replace the symbols whose meaning changed in round 4; share format 2);
 * 6 = round 6 (additions only: sonic_prove_batch fuses the proofs of a handle group, sonic_one_shot_trim) */
int sonic_abi_version(void);
int sonic_init(int device_ordinal);                 /* choose the DEFAULT GPU (first call wins) and make it the thread's HIP device; idempotent */
int sonic_device_count(int* out);                   /* GPUs this process can see; SONIC_ERR_NO_DEVICE (and 0) without one */
int sonic_last_error(char* buf, size_t cap);        /* copies the calling thread's last message */
int sonic_device_sync(void);
/* HIP_VERSION the library was compiled against / hipRuntimeGetVersion of the runtime mapped into this process (major * 10^7 +
 * minor * 10^5 + patch).  A process holds ONE HIP runtime; when another component loaded its own first (a PyTorch-ROCm wheel), the
 * two differ and the caller may want to know.  No device needed. */
int sonic_hip_versions(int* build, int* runtime);

/* ---- Sonic.SRS ---- */
/* SRS.new :: Int -> Fr -> Fr -> SRS  (SRS.hs:27-43).  Generates, on the GPU, the G1 halves the
 * prover reads: basis 0 = g^{x^e}, basis 1 = g^{alpha x^e}, e in [-d, d]; slot e = 0 of basis 1 is
 * empty (g^alpha is not shared, SRS.hs:38).  gNegativeX[k] = basis0[-(k+1)], gPositiveX[k] =
 * basis0[k], gNegativeAlphaX[k] = basis1[-(k+1)], gPositiveAlphaX[k] = basis1[k+1]. */
int sonic_srs_new(int64_t d, const uint8_t x[32], const uint8_t alpha[32], sonic_srs_t** out);
/* the same on a named GPU (SURVEY 8b proposed `sonic_srs_new(d, x, alpha, n_gpus)`: one replica per device, made by one call each --
 * the calls may run on different host threads at once).  device < 0: the default device. */
int sonic_srs_new_on(int device, int64_t d, const uint8_t x[32], const uint8_t alpha[32], sonic_srs_t** out);
/* the record constructor `SRS{..}`: caller-supplied points, (2d+1) * 96 bytes per basis.  Every point is checked to be
 * canonical, on the curve and in the prime-order subgroup G1 (r P = O, as the powers of a generator are): MSMs over an
 * SRS rely on it (scalars above r/2 run as r - s on the negated point).  SONIC_ERR_BAD_ENCODING otherwise. */
int sonic_srs_from_points(int64_t d, const uint8_t* basis0, const uint8_t* basis1, sonic_srs_t** out);
int sonic_srs_from_points_on(int device, int64_t d, const uint8_t* basis0, const uint8_t* basis1, sonic_srs_t** out);
/* a replica of `srs` on another GPU of this process: the G1 bases and their window tables are copied device to device
 * (hipMemcpyPeer; no re-validation, no table rebuild), the G2 half if the handle holds it; a handle that could still generate its
 * G2 half hands its trapdoor on to the replica as well.  The replica runs the same MSM plans as the original (same tables), which
 * sonic_prove_shared relies on. */
int sonic_srs_replicate(const sonic_srs_t* srs, int device, sonic_srs_t** out);
void sonic_srs_free(sonic_srs_t* srs);
int64_t sonic_srs_d(const sonic_srs_t* srs);        /* srsD */
int sonic_srs_device(const sonic_srs_t* srs);       /* the GPU the handle lives on (-1 for NULL) */
/* srsPairing = e(g, h^alpha) (SRS.hs:21,42), the one record field that is neither a vector nor d: 576 bytes, the Fq12 element as
 * its twelve Fq coefficients c[i][j][k] (i: the Fq6 half of Fq12 = Fq6[w]/(w^2 - v), j: the Fq2 coefficient of Fq6 = Fq2[v]/(v^3 - (1 + u)),
 * k: real / imaginary part), 48-byte little-endian each -- the layout of oracle/pairing.py's nested tuples flattened.  Host pairing
 * (pairing.hpp) of the generator with hPositiveAlphaX[0]; needs the G2 half. */
int sonic_srs_pairing(const sonic_srs_t* srs, uint8_t out[576]);
/* basis: 0 = g^{x^e}, 1 = g^{alpha x^e} (the four reference vectors, SRS.hs:33-39); diagnostics: b + 2w = window table w of basis b,
 * SONIC_BASIS_ALPHA_PREFIX = the running sums of the alpha basis, entry e = sum of g^{alpha x^k} over k in [-d, e] (what a run of equal
 * coefficients is committed with: DESIGN.md section 4),
 * SONIC_BASIS_ALPHA_SYM = the symmetric sums g^{alpha x^e} + g^{alpha x^-e}, e in [1, d] (what hscProve's C is committed with) */
#define SONIC_BASIS_ALPHA_PREFIX 1000
#define SONIC_BASIS_ALPHA_SYM 1001
int sonic_srs_get_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out);
/* the G2 half (SRS.hs:35-36,40-41): basis 0 = h^{x^e}, basis 1 = h^{alpha x^e}, e in [-d, d]; hNegativeX[k] = basis0[-(k+1)],
 * hPositiveX[k] = basis0[k], hPositiveAlphaX[k] = basis1[k], hNegativeAlphaX[k] = basis1[-(k+1)].  G2 encoding: 192 bytes
 * x.c0 || x.c1 || y.c0 || y.c1 (48-byte little-endian each, x = c0 + c1 u), infinity = zeros.  A handle made by
 * sonic_srs_new generates it on the GPU on first use and then forgets x and alpha; other handles have it only after
 * sonic_srs_set_g2_points or a load from a file that carries it (SONIC_ERR_INVALID_ARG otherwise). */
int sonic_srs_get_g2_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out);
/* attaches caller-supplied G2 vectors, (2d+1) * 192 bytes per basis, checked like the G1 side (canonical, on the twist,
 * r P = O); the prover never reads them, the verifier reads three elements (CommitmentScheme.hs:58-68). */
int sonic_srs_set_g2_points(sonic_srs_t* srs, const uint8_t* basis0, const uint8_t* basis1);
/* on-disk SRS (the reference has no persistence): "SONICSRS", u32 version = 2, u32 flags (bit 0: G2 half present), i64 d,
 * the two G1 bases as (2d+1) x 96 canonical bytes each, then -- with_g2 != 0 -- the two G2 bases as (2d+1) x 192 bytes
 * each.  Loading validates every point like sonic_srs_from_points / sonic_srs_set_g2_points; version-1 files (G1 only)
 * still load.  A file never holds x or alpha.  with_g2: 0 = G1 only, 1 = with the G2 half (SONIC_ERR_INVALID_ARG if the handle
 * has none), 2 = with the G2 half if the handle has it (sonic_srs_has_g2).  (ABI note: the with_g2 argument was added in
 * round 2; callers built against the two-argument prototype must be recompiled.)  Infinity is rejected in every SRS input
 * (from_points except the omitted g^alpha slot, set_g2_points, load): no power of a generator is the identity. */
int sonic_srs_save(const sonic_srs_t* srs, const char* path, int with_g2);
/* 1 if the handle holds the G2 half or can still generate it (made by sonic_srs_new and not yet used), else 0 */
int sonic_srs_has_g2(const sonic_srs_t* srs);
int sonic_srs_load(const char* path, sonic_srs_t** out);
int sonic_srs_load_on(int device, const char* path, sonic_srs_t** out);

/* ---- Sonic.CommitmentScheme ---- */
/* commitPoly :: SRS -> Int -> VLaurent Fr -> G1  (CommitmentScheme.hs:20-33) */
int sonic_commit_poly(const sonic_srs_t* srs, int64_t max, int64_t n_terms, const int64_t* exps,
                      const uint8_t* coeffs, uint8_t out_g1[96]);
/* openPoly :: SRS -> Fr -> VLaurent Fr -> (Fr, G1)  (CommitmentScheme.hs:36-48) */
int sonic_open_poly(const sonic_srs_t* srs, const uint8_t z[32], int64_t n_terms, const int64_t* exps,
                    const uint8_t* coeffs, uint8_t out_fz[32], uint8_t out_g1[96]);

/* ---- the kernels behind them, exposed for parity tests and the MSM benchmark ---- */
/* foldl' (\acc (P, v) -> acc <> P `mul` v) mempty  (the fold at CommitmentScheme.hs:26-29, 45-48).  Points only have to be
 * on the curve: s P is the literal multiple (no use of r P = O), also for cofactor points such as (0, 2). */
int sonic_msm_g1(const uint8_t* points, const uint8_t* scalars, int64_t n, uint8_t out_g1[96]);
/* same over an SRS slice e0 .. e0+n-1 of one basis; scalars on the host */
int sonic_msm_g1_srs(const sonic_srs_t* srs, int basis, int64_t e0, const uint8_t* scalars, int64_t n,
                     uint8_t out_g1[96]);
/* scalars already resident in HBM (canonical 32-byte Fr, device pointer) */
int sonic_msm_g1_srs_dev(const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n,
                         uint8_t out_g1[96]);
/* one rank's share of a range-sharded MSM: the un-normalised partial sum */
int sonic_msm_g1_srs_partial_dev(const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars,
                                 int64_t n, uint8_t out_partial[192]);
/* the same fold (CommitmentScheme.hs:26-29, 45-48 over an SRS slice, as sonic_msm_g1_srs_dev) in two halves on a lane of its
 * own (stream, bucket workspace, pinned result): submit queues it and returns,
 * collect waits and finishes it (out_g1: 96 canonical bytes, out_partial: 192-byte un-normalised sum; either may be NULL).
 * One MSM in flight per lane; two lanes used in turn stream MSMs with the sort and the reduction of one under the
 * accumulation of the other.  d_scalars: canonical 32-byte Fr resident in HBM, untouched until collect. */
typedef struct sonic_msm_lane sonic_msm_lane_t;
int sonic_msm_lane_new(sonic_msm_lane_t** out);
int sonic_msm_lane_new_on(int device, sonic_msm_lane_t** out);      /* a lane serves SRS handles of its own GPU (SONIC_ERR_INVALID_ARG otherwise) */
void sonic_msm_lane_free(sonic_msm_lane_t* lane);
int sonic_msm_submit(sonic_msm_lane_t* lane, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n);
int sonic_msm_collect(sonic_msm_lane_t* lane, uint8_t* out_g1, uint8_t* out_partial);
/* a lane on a stream the caller owns (e.g. the stream its RCCL collectives are ordered on); the stream must outlive the lane */
int sonic_msm_lane_new_on_stream(void* hip_stream, sonic_msm_lane_t** out);
/* sonic_msm_submit that also leaves the un-normalised result (SONIC_G1_DEV_PARTIAL_BYTES) in device memory, queued on the lane's stream:
 * the operand of a cross-rank all-gather that never visits the host.  out_bytes: the size of d_partial_out, checked against
 * SONIC_G1_DEV_PARTIAL_BYTES.  (sonic_msm_submit_dev, which wrote 192 bytes up to round 3 and 12304 in round 4 under one name, is
 * retired: the symbol still links and returns SONIC_ERR_INVALID_ARG.) */
int sonic_msm_submit_dev_v2(sonic_msm_lane_t* lane, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n,
                            void* d_partial_out, size_t out_bytes);
int sonic_msm_submit_dev(sonic_msm_lane_t* lane, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n,
                         void* d_partial_out);      /* retired: SONIC_ERR_INVALID_ARG */
/* ONE MSM strong-scaled over `world` GPUs by sharding its BUCKETS (the fold of CommitmentScheme.hs:25-29 / 45-48 split twice: by term
 * range for the accumulation, by bucket range for the reduction):
 *   layout      n_buckets of the shared bucket set and the slice length S (a multiple of 16384, world * S >= n_buckets)
 *   accumulate  sort + bucket accumulation of this rank's terms into d_buckets (capacity >= world * S entries of 192 B; the
 *               padding is cleared), queued on the lane's stream
 *   -- the caller exchanges slices: all-to-all, rank r receives entries [r S, (r+1) S) of every rank, laid out [world][S] --
 *   reduce      element-wise curve addition of the k slices, then sum_i (bucket_base + i + 1) * slice[i]: SONIC_G1_DEV_PARTIAL_BYTES in
 *               device memory (gathered and added like the term-range partials), queued on the lane's stream
 *   sync        waits for the lane, reports a non-canonical scalar */
int sonic_msm_exchange_layout(const sonic_srs_t* srs, int world, int64_t* n_buckets, int64_t* slice_len);
int sonic_msm_accumulate_dev(sonic_msm_lane_t* lane, const sonic_srs_t* srs, int basis, int64_t e0, const void* d_scalars, int64_t n,
                             void* d_buckets, int64_t capacity);
int sonic_msm_reduce_slices_dev_v2(sonic_msm_lane_t* lane, const sonic_srs_t* srs, const void* d_slices, int k, int64_t slice_len,
                                   int64_t bucket_base, void* d_partial_out, size_t out_bytes);
int sonic_msm_reduce_slices_dev(sonic_msm_lane_t* lane, const sonic_srs_t* srs, const void* d_slices, int k, int64_t slice_len,
                                int64_t bucket_base, void* d_partial_out);      /* retired like sonic_msm_submit_dev: SONIC_ERR_INVALID_ARG */
int sonic_msm_lane_sync(sonic_msm_lane_t* lane);
/* curve addition of k partials (RCCL has no such reduction op) + normalisation */
int sonic_g1_sum_partials(const uint8_t* partials, int k, uint8_t out_g1[96]);
/* the same for k device-side results of SONIC_G1_DEV_PARTIAL_BYTES each (host only) */
int sonic_g1_sum_dev_partials(const uint8_t* blobs, int k, uint8_t out_g1[96]);
/* in-place radix-2 NTT over Fr, natural order in and out; omega = 7^((r-1)/2^log2n); 0 <= log2n <= 28 (SONIC_ERR_INVALID_ARG beyond) */
int sonic_ntt_fr(uint8_t* data, int log2n, int inverse);
/* dense product of two coefficient arrays (the `*` at Constraints.hs:61): out has na+nb-1 Fr */
int sonic_poly_mul_fr(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb, uint8_t* out);
/* the same with operands and result resident in HBM (device pointers to canonical Fr; d_out: na + nb - 1 elements) */
int sonic_poly_mul_fr_dev(const void* d_a, int64_t na, const void* d_b, int64_t nb, void* d_out);
/* bytes between consecutive points of an SRS basis / window table in HBM (128: one HBM line per 96-byte point); what a bucket walk
 * reads per (term, window) by design */
int sonic_srs_point_bytes(void);
/* MSM tuning knob for tests: window bits (0 = automatic) */
int sonic_msm_set_window(int c);
/* what an n-term MSM over this SRS will run as: window bits, number of windows, and bucket sets
 * (1 = all windows share one bucket set over the precomputed window tables, else one set per window) */
int sonic_msm_plan(const sonic_srs_t* srs, int64_t n, int* window_bits, int* windows, int* bucket_sets);

/* ---- Sonic.Protocol / Sonic.Signature ---- */
size_t sonic_proof_size(int64_t Q);
/* prove :: SRS -> Assignment Fr -> ArithCircuit Fr -> m (Proof, RndOracle)  (Protocol.hs:47-109),
 * including hscProve (Signature.hs:38-72).  n = length aL, Q = length wL. */
int sonic_prove(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR,
                const uint8_t* wO, const uint8_t* cs, const uint8_t* aL, const uint8_t* aR,
                const uint8_t* aO, const uint8_t* transcript, uint8_t* out_proof);
/* sonic_prove parks the shell of a finished call (streams, workspaces of a proof of that shape, twiddle tables: several GB at n = 2^20)
 * for the next call with the same SRS handle and (n, Q); at most four per GPU stay parked, and freeing an SRS frees the shells over it.
 * sonic_one_shot_trim frees the parked shells now -- of one GPU, or of all (device < 0) -- and returns how many there were. */
int sonic_one_shot_trim(int device);
/* the same split so that circuit and assignment stay resident in HBM across proofs */
int sonic_prover_new(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR,
                     const uint8_t* wO, const uint8_t* cs, sonic_prover_t** out);
int sonic_prover_set_assignment(sonic_prover_t* p, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO);
/* optional, once per handle: commits the Q constraint-row polynomials of sPoly (Constraints.hs:34-53), after which every
 * S_j = commitPoly(s(X, y_j)) (Signature.hs:42) costs an n-term MSM instead of a 3n-term one.  Same proof bytes. */
int sonic_prover_prepare(sonic_prover_t* p);
int sonic_prover_prove(sonic_prover_t* p, const uint8_t* transcript, uint8_t* out_proof);
/* prove (Protocol.hs:47-109) in two halves: submit queues the whole proof on the handle's streams and returns without waiting for the GPU;
 * collect waits for it, finishes it on the host and writes the bytes (prove = submit + collect).  One proof in flight per
 * handle.  A host thread that alternates between two handles -- submit(A, t0), submit(B, t1), collect(A), submit(A, t2),
 * collect(B), ... -- streams proofs with the next proof's start-up under the previous proof's tail (the reference's
 * `mapM prove` over a list of statements; BASELINE config "batch of independent proofs, throughput mode"). */
int sonic_prover_submit(sonic_prover_t* p, const uint8_t* transcript);
int sonic_prover_collect(sonic_prover_t* p, uint8_t* out_proof);
void sonic_prover_free(sonic_prover_t* p);

/* ---- ONE proof over several GPUs ----
 * The 7 + 4Q commitments and openings of prove + hscProve (Protocol.hs:63,73,79-81; Signature.hs:40-45,51-57,63) are independent
 * sums once the transcript is known.  Every rank (one process per GPU) makes a handle for the same circuit and assignment over its
 * replica of the SRS and calls set_share(rank, world): the handle then builds only the polynomials its pieces read and runs a
 * contiguous piece of the proof's MSMs laid end to end (a cut inside an MSM splits its term range), balanced by a cost model
 * (sonic_amd/csrc/share_plan.hpp).  submit + collect_share (or prove_share) with the SAME transcript on every rank yield one share
 * per rank -- un-normalised 192-byte partial sums per MSM, the evaluations the rank computed, its error flags: sonic_proof_share_size
 * bytes; the caller all-gathers them (RCCL / any transport) and sonic_proof_from_shares lays out the proof, byte-identical to
 * sonic_prover_prove on one GPU.  world <= 1 restores the whole proof.  Not available for sonic_prover_prove_fs (the Fiat-Shamir
 * chain serialises the MSMs). */
int sonic_prover_set_share(sonic_prover_t* p, int rank, int world);
size_t sonic_proof_share_size(int64_t Q);
int sonic_prover_prove_share(sonic_prover_t* p, const uint8_t* transcript, uint8_t* out_share);
int sonic_prover_collect_share(sonic_prover_t* p, uint8_t* out_share);       /* after sonic_prover_submit */
/* shares: world x sonic_proof_share_size(Q) bytes in any rank order; host only.  Fails with SONIC_ERR_INVALID_ARG unless the pieces
 * of every MSM cover its terms exactly once and every evaluation is reported; a rank's error flags become the status sonic_prove
 * would have returned. */
int sonic_proof_from_shares(int64_t Q, int world, const uint8_t* shares, const uint8_t* transcript, uint8_t* out_proof);
/* the plan itself (host only): out_lo_hi = 7 + 4Q pairs {lo, hi} in units of 1 / 2^20 of each MSM's terms, slot order R, T, W_a,
 * W_b, W_t, [S_j, W_j]_j, [W'_j, Q_j]_j, Q_v, C; out_cost (may be NULL): the rank's modelled cost in MSM terms.  nb, w: buckets per
 * set and windows of the MSM plan (0, 0: 2^19 and 13, an SRS with window tables at d >= 2^21). */
int sonic_prove_share_plan(int64_t n, int64_t Q, int prepared, int world, int rank, int64_t nb, int w, uint32_t* out_lo_hi, double* out_cost);

/* ---- N GPUs from ONE host process (round 5) ----
 * The reference's prove is one pure call in one process (Protocol.hs:47-52); these keep it that way on a node of GPUs.  Each takes an
 * array of handles that live on the GPUs to use (one SRS replica per device: sonic_srs_new_on / sonic_srs_replicate; the same device
 * may appear more than once, which is how the tests drive them on a one-GPU box) and runs one host thread per handle inside the
 * library.  Nothing here needs torch, RCCL or a launcher; sonic_amd/distributed.py keeps the one-process-per-GPU form over RCCL.
 *
 *   sonic_prove_shared    ONE proof over `world` prover handles of the same circuit and assignment: handle r runs rank r's share
 *                         (sonic_prover_set_share is applied by the call), the 3.3-KB shares are combined on the host
 *                         (sonic_proof_from_shares).  Same bytes as sonic_prover_prove on one GPU.
 *   sonic_prove_batch     K proofs of ONE circuit over n_provers handles (`mapM (prove srs asg_i circuit)`; for K statements with their own
 *                         circuits see sonic_prove_many): proof i goes to handle i % n_provers, each handle proves its share of the
 *                         list in order on a thread of its own; no collective.  Two handles per GPU stream that GPU's proofs (one
 *                         proof's host tail under the next proof's kernels).  aL, aR, aO: K x n x 32 bytes each, or all NULL to keep
 *                         every handle's resident assignment; transcripts: K x (8 + 2Q) x 32; out_proofs: K x sonic_proof_size(Q);
 *                         out_status (may be NULL): K statuses.  Returns the first non-zero status in list order (all proofs are
 *                         attempted).
 *   sonic_msm_g1_srs_multi        ONE MSM sum_i s_i B[e0 + i] over `world` SRS replicas, scalars on the host (n x 32 bytes);
 *   sonic_msm_g1_srs_multi_dev    the same with rank r's slice resident on srs[r]'s GPU: e0[r], d_scalars[r], n[r].
 *                         mode 0 = by term range: every GPU runs a whole MSM over its slice, the host adds the `world` partial sums
 *                         (CommitmentScheme.hs:25-29 is a fold over independent terms); mode 1 = by bucket range (strong scaling):
 *                         every GPU accumulates its slice into a full bucket set, the GPUs pull their bucket range from each other
 *                         (hipMemcpyPeerAsync: the all-to-all of sonic_msm_exchange_layout), each reduces 1/world of the buckets;
 *                         needs the full window tables on every replica.
 *   sonic_prove_many      K INDEPENDENT statements of one shape (n, Q) -- every proof its own circuit, assignment and transcript, as host
 *                         buffers: `mapM (uncurry (prove srs))` -- spread over the SRS replicas with two host threads per replica, each
 *                         making one-shot sonic_prove calls (the device parks the handles' shells between calls).  out_proofs: K x
 *                         sonic_proof_size(Q); out_status (may be NULL): K statuses; returns the first non-zero one in list order. */
typedef struct sonic_statement {
  const uint8_t *wL, *wR, *wO, *cs;      /* ArithCircuit: Q x n weights each, Q constants */
  const uint8_t *aL, *aR, *aO;           /* Assignment: n each */
  const uint8_t* transcript;             /* 8 + 2Q draws */
} sonic_statement_t;
int sonic_prove_many(const sonic_srs_t* const* srs, int n_srs, int64_t n, int64_t Q, const sonic_statement_t* statements, int64_t K,
                     uint8_t* out_proofs, int* out_status);
int sonic_prove_shared(sonic_prover_t* const* provers, int world, const uint8_t* transcript, uint8_t* out_proof);
int sonic_prove_batch(sonic_prover_t* const* provers, int n_provers, int64_t K, const uint8_t* aL, const uint8_t* aR, const uint8_t* aO,
                      const uint8_t* transcripts, uint8_t* out_proofs, int* out_status);
int sonic_msm_g1_srs_multi(const sonic_srs_t* const* srs, int world, int basis, int64_t e0, const uint8_t* scalars, int64_t n, int mode,
                           uint8_t out_g1[96]);
int sonic_msm_g1_srs_multi_dev(const sonic_srs_t* const* srs, int world, int basis, const int64_t* e0, const void* const* d_scalars,
                               const int64_t* n, int mode, uint8_t out_g1[96]);
int sonic_prover_device(const sonic_prover_t* p);   /* the GPU a prover handle lives on (its SRS's); -1 for NULL */

/* ---- opt-in Fiat-Shamir transcript ----
 * The reference draws its challenges with `rnd` (Protocol.hs:58,66,76,84-85; Signature.hs:48,60) and hands y, z, (y_j, z_j) to the
 * verifier as RndOracle.  In this mode each draw is instead SHA-256 of everything that precedes it, in that order (exact
 * definition: sonic_amd/csrc/fs.hpp): the proof carries its own challenges.  It serialises the
 * proof (R -> y -> T -> z -> ...: six waits for the GPU instead of one), so the explicit transcript above stays the default.
 *   circuit_digest  SHA-256 of (n, Q, wL, wR, wO, cs): computed once per circuit
 *   srs_id          SHA-256 of d and the four G1 elements g^x, g^{alpha x}, g^{1/x}, g^{alpha/x}: binds the transcript to ONE reference
 *                   string (round 4; the round-3 transcript bound only d)
 *   prove_fs        blinder_seed: the prover's secret randomness (32 bytes).  The four blinders are derived from the seed AND the
 *                   circuit digest, the srs id and a digest of the assignment (RFC 6979 style, round 4), so one seed may serve
 *                   several statements; a seed must still never be disclosed;
 *                   out_transcript (may be NULL): the 8 + 2Q values the proof was made with, in sonic_prove's transcript order --
 *                   sonic_prover_prove on them reproduces the proof byte for byte
 *   fs_challenges   what a proof determines: y, z, y_1..y_Q, z_1..z_Q, u, v (32 bytes each)
 *   verify_fs       recomputes them, requires the proof's own u, v to match, then verify (Protocol.hs:111-130) */
int sonic_fs_circuit_digest(int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO, const uint8_t* cs, uint8_t out[32]);
int sonic_prover_prove_fs(sonic_prover_t* p, const uint8_t circuit_digest[32], const uint8_t blinder_seed[32], uint8_t* out_proof,
                          uint8_t* out_transcript);
int sonic_fs_srs_id(const sonic_srs_t* srs, uint8_t out[32]);
int sonic_fs_challenges_v2(int64_t n, int64_t Q, int64_t d, const uint8_t circuit_digest[32], const uint8_t srs_id[32], const uint8_t* proof, uint8_t* out);
/* retired (round 3's prototype, without srs_id; round 4 changed the argument list under this name): SONIC_ERR_INVALID_ARG */
int sonic_fs_challenges(int64_t n, int64_t Q, int64_t d, const uint8_t circuit_digest[32], const uint8_t* proof, uint8_t* out);
int sonic_verify_fs(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                    const uint8_t* cs, const uint8_t* proof, int* accepted);

/* hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof  (Signature.hs:32-72) on its own, for the s(X,Y) of the handle's
 * circuit (Constraints.hs:34-53) and any number m of (y_j, z_j) pairs (yzs: m x 64 bytes); u, v are its two `rnd` draws.
 * out (sonic_hsc_proof_size(m) bytes): [S_j, s_j, W_j]_j, [s'_j, W'_j, Q_j]_j, Q_v, C, u, v -- the HscProof part of a proof. */
size_t sonic_hsc_proof_size(int64_t m);
int sonic_prover_hsc_prove(sonic_prover_t* p, int64_t m, const uint8_t* yzs, const uint8_t u[32], const uint8_t v[32], uint8_t* out);

/* hscProve with the reference's own signature (Signature.hs:32-37): ANY sparse bivariate Laurent polynomial
 * s(X,Y) = sum_i coeffs[i] X^{x_exps[i]} Y^{y_exps[i]} (terms in any order, repeated exponent pairs are summed), m pairs (y_j, z_j),
 * the two `rnd` draws u, v.  Same output layout as sonic_prover_hsc_prove.  An evaluation point may be zero only if the
 * polynomial has no negative power of that variable (`pow 0 e`, e < 0, divides by zero in the reference). */
int sonic_hsc_prove_poly(const sonic_srs_t* srs, int64_t n_terms, const int64_t* x_exps, const int64_t* y_exps, const uint8_t* coeffs,
                         int64_t m, const uint8_t* yzs, const uint8_t u[32], const uint8_t v[32], uint8_t* out);

/* ---- the verifier side of the API (host CPU; outside the accelerated path) ---- */
/* pcV :: SRS -> Int -> G1 -> Fr -> (Fr, G1) -> Bool  (CommitmentScheme.hs:51-68); *accepted = 0 / 1 */
int sonic_pc_v(const sonic_srs_t* srs, int64_t max, const uint8_t commitment[96], const uint8_t z[32], const uint8_t v[32],
               const uint8_t w[96], int* accepted);
/* verify :: SRS -> ArithCircuit Fr -> Proof -> Fr -> Fr -> [(Fr, Fr)] -> Bool  (Protocol.hs:111-130), including
 * hscVerify (Signature.hs:74-90).  yzs = Q pairs y_j || z_j (64 bytes each), i.e. rndOracleYZs. */
int sonic_verify(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                 const uint8_t* cs, const uint8_t* proof, const uint8_t y[32], const uint8_t z[32], const uint8_t* yzs, int* accepted);

/* hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool  (Signature.hs:74-90) for the s(X,Y) of a circuit */
int sonic_hsc_verify(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                     int64_t m, const uint8_t* yzs, const uint8_t* hsc, int* accepted);

/* hscVerify (Signature.hs:74-90) for any sparse bivariate Laurent polynomial, the counterpart of sonic_hsc_prove_poly */
int sonic_hsc_verify_poly(const sonic_srs_t* srs, int64_t n_terms, const int64_t* x_exps, const int64_t* y_exps, const uint8_t* coeffs,
                          int64_t m, const uint8_t* yzs, const uint8_t* hsc, int* accepted);

/* ---- device memory for callers without a HIP binding ---- */
int sonic_dev_alloc(size_t bytes, void** out);                       /* on the default device */
int sonic_dev_alloc_on(int device, size_t bytes, void** out);         /* free / upload / download find the pointer's device themselves */
int sonic_dev_free(void* p);
int sonic_dev_upload(void* dst, const void* src, size_t bytes);
int sonic_dev_download(void* dst, const void* src, size_t bytes);

/* ---- per-kernel HIP-event timing (bench.py's roofline leg) ---- */
int sonic_profile_enable(int on);
int sonic_profile_reset(void);
int sonic_profile_get(const char* kernel, double* total_ms, int64_t* launches);
int sonic_profile_names(char* buf, size_t cap);     /* newline-separated kernel names seen so far */

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif
