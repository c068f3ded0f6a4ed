// Prover-specific polynomial builders (poly.hip); see internal.hpp for the generic ones.
#pragma once
#include "common.hpp"
#include "field.hpp"
#include "g1.hpp"

namespace sonic {

void build_r1_enqueue(hipStream_t st, const Fr* aL, const Fr* aR, const Fr* aO, const Fr* cns, long n, Fr* r1);
void s_of_y_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, const Fr* ypow, long n, long Q, Fr* s);
void weight_row_poly_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, long n, long q, Fr* s);
void s_diag_part_enqueue(hipStream_t st, const Fr* ypow, long n, long Q, Fr* diag, Fr* yq);
void s_of_u_enqueue(hipStream_t st, const Fr* wL, const Fr* wR, const Fr* wO, const Fr* upow, long n, long Q, Fr* s, DevBuf& tmp);
void add_into_enqueue(hipStream_t st, Fr* dst, const Fr* src, long n);
void t_operands_enqueue(hipStream_t st, const Fr* r1, long r_len, long r_lo, const Fr* sy, long s_off, long s_len, const Fr* ypair, Fr* fa, Fr* fb, long M);
void sub_k_of_y_enqueue(hipStream_t st, Fr* slot, const Fr* cs, const Fr* ypow_nq, long Q, int* flags, int flag_bit);
void flag_nonzero_enqueue(hipStream_t st, const Fr* a, long n, int* flags, int bit);
// runs of equal coefficients (poly.hip): tiles of RUN_TILE coefficients that hold one non-zero value are zeroed in `masked` and recorded;
// run_terms turns the records into 2 * (len / RUN_TILE) (scalar, running-sum point) slots
constexpr int RUN_TILE = 256;
void run_tiles_enqueue(hipStream_t st, const Fr* poly, long len, Fr* masked, Fr* val, uint32_t* uniform);
void run_terms_enqueue(hipStream_t st, const Fr* val, const uint32_t* uniform, long ntiles, PointArray ps, long ps_first, Fr* scal, G1Affine* pts);
void fr_with_inverse_enqueue(hipStream_t st, const Fr* in, int k, Fr* out);
void fr_mul_scalar_enqueue(hipStream_t st, const Fr* a, const Fr* b, Fr* out);
void scale_terms_enqueue(hipStream_t st, const int64_t* e, const Fr* c, long nt, const Fr* pair, Fr* out);
void sparse_to_dense_enqueue(hipStream_t st, const int64_t* exps, const Fr* coeffs, long nt, long lo, Fr* dense);

// Several openings of polynomials over ONE exponent range [lo, lo + len) as one launch per step (round 6): D_k[i] = poly_k[i] z_k^(lo + i),
// inclusive prefix sums of every D_k, fz_k = the last prefix, q_k = the quotient (f_k(X) - f_k(z_k)) / (X - z_k) over [lo, lo + len - 2].
// What poly_scale_powers / poly_prefix_sum / poly_quotient do for one opening in six launches, for up to OPEN_BATCH_MAX openings in
// five: the openings of a group (r(X,1) at z and yz; s(X,y_j) at z_j and u; s(u,Y) at y_1 .. y_Q and v) share their polynomial.
constexpr int OPEN_BATCH_MAX = 16;
struct OpenBatch {
  int k;
  const Fr* poly[OPEN_BATCH_MAX];
  Fr* D[OPEN_BATCH_MAX];            // len + 1 entries each
  Fr* q[OPEN_BATCH_MAX];            // len + 1 entries each
  Fr* tiles[OPEN_BATCH_MAX];        // ceil(len / 1024) + 1 entries each
  const Fr* zpair[OPEN_BATCH_MAX];  // {z, z^-1}
  Fr* fz[OPEN_BATCH_MAX];           // where f(z) goes (never null)
};
void open_batch_enqueue(hipStream_t st, const OpenBatch& b, long lo, long len, bool quotient = true);      // quotient = false: only the evaluations fz_k

}  // namespace sonic
