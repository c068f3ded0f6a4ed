// The C-ABI faces of the reference's polynomial-level functions on CALLER-SUPPLIED polynomials -- hscProve for any sparse bivariate Laurent
// polynomial (src/Sonic/Signature.hs:32-72), commitPoly / openPoly (src/Sonic/CommitmentScheme.hs:20-40) -- and the stand-alone NTT / dense
// product (the `*` at src/Sonic/Constraints.hs:61).  The same kernels and MSM groups as inside prove() (prove.hip), one call at a time on the
// device's default stream.
#include "prover.hpp"

extern "C" {

// hscProve :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> m HscProof (Signature.hs:32-72) for ANY sparse bivariate Laurent polynomial
// s(X,Y) = sum_i c_i X^{ex_i} Y^{ey_i} (the reference's own signature; sonic_prover_hsc_prove above is the same sub-protocol for the
// s(X,Y) of a circuit handle).  evalY y_j / evalX u (Utils.hs:17-21) scale every term by a power of the evaluation point and sum
// the terms that share the remaining exponent; the commitments and openings are the usual MSM groups.
struct BivTerms {
  long nt = 0, lo = 0, len = 0;       // dense range of the variable that is kept
  DevBuf keep, other, coeff;          // per term, sorted by the kept exponent: kept exponent, substituted exponent, coefficient (Montgomery)
};
static int biv_upload(hipStream_t st, int64_t nt, const int64_t* keep, const int64_t* other, const uint8_t* coeffs, BivTerms& out, int* d_flags) {
  out.nt = nt;
  long lo = 0, hi = 0;                // the range always holds exponent 0 (openPoly puts -f(z) there, CommitmentScheme.hs:43)
  for (int64_t i = 0; i < nt; i++) { lo = std::min<long>(lo, keep[i]); hi = std::max<long>(hi, keep[i]); }
  out.lo = lo; out.len = hi - lo + 1;
  std::vector<int64_t> order(nt), k(nt), o(nt);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return keep[a] < keep[b]; });
  std::vector<uint8_t> c(32 * (size_t)(nt ? nt : 1));
  for (int64_t i = 0; i < nt; i++) { k[i] = keep[order[i]]; o[i] = other[order[i]]; memcpy(&c[32 * (size_t)i], coeffs + 32 * order[i], 32); }
  out.keep.alloc(8 * (size_t)(nt ? nt : 1)); out.other.alloc(8 * (size_t)(nt ? nt : 1)); out.coeff.alloc(32 * (size_t)(nt ? nt : 1));
  if (nt) {
    HIP_OK(hipMemcpyAsync(out.keep.p, k.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(out.other.p, o.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(out.coeff.p, c.data(), 32 * (size_t)nt, hipMemcpyHostToDevice, st));
    fr_to_mont_enqueue(st, out.coeff.as<Fr>(), nt, d_flags);
  }
  HIP_OK(hipStreamSynchronize(st));   // host staging goes out of scope
  return SONIC_OK;
}
// dense[e - lo] = sum over the terms with kept exponent e of c * b^{other exponent}
static void biv_eval_enqueue(hipStream_t st, const BivTerms& t, const Fr* pair, DevBuf& scaled, Fr* dense) {
  HIP_OK(hipMemsetAsync(dense, 0, sizeof(Fr) * t.len, st));
  scaled.ensure(sizeof(Fr) * (size_t)(t.nt ? t.nt : 1));
  scale_terms_enqueue(st, t.other.as<int64_t>(), t.coeff.as<Fr>(), t.nt, pair, scaled.as<Fr>());
  sparse_to_dense_enqueue(st, t.keep.as<int64_t>(), scaled.as<Fr>(), t.nt, t.lo, dense);
}

int sonic_hsc_prove_poly(const sonic_srs_t* srs, int64_t n_terms, const int64_t* x_exps, const int64_t* y_exps, const uint8_t* coeffs,
                         int64_t m, const uint8_t* yzs, const uint8_t u[32], const uint8_t v[32], uint8_t* out) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || n_terms < 0 || (n_terms > 0 && (!x_exps || !y_exps || !coeffs)) || m < 0 || (m > 0 && !yzs) || !u || !v || !out) return SONIC_ERR_INVALID_ARG;
  CallLease lease;
  hipStream_t st = lease.st();
  const long d = srs_d(srs);
  bool neg_x = false, neg_y = false;
  for (int64_t i = 0; i < n_terms; i++) {
    neg_x = neg_x || x_exps[i] < 0; neg_y = neg_y || y_exps[i] < 0;
    if (x_exps[i] < -8 * (d + 8) || x_exps[i] > 8 * (d + 8) || y_exps[i] < -8 * (d + 8) || y_exps[i] > 8 * (d + 8)) {
      set_error("hscProve: exponent (%ld, %ld) is far outside the SRS", (long)x_exps[i], (long)y_exps[i]); return SONIC_ERR_SRS_INDEX; }
  }
  // `pow x e` with negative e (Utils.hs:18,21): substituting 0 divides by zero; so does opening at 0 a polynomial with negative exponents
  for (int64_t j = 0; j < m; j++) {
    if (bytes_are_zero(yzs + 64 * j, 32) && neg_y) { set_error("hscProve: y_%ld = 0 and s(X,Y) has negative powers of Y", (long)j); return SONIC_ERR_INEXACT_DIVISION; }
    if (bytes_are_zero(yzs + 64 * j + 32, 32) && neg_x) { set_error("hscProve: z_%ld = 0 and s(X,Y) has negative powers of X", (long)j); return SONIC_ERR_INEXACT_DIVISION; }
  }
  if ((bytes_are_zero(u, 32) && neg_x) || (bytes_are_zero(v, 32) && neg_y)) { set_error("hscProve: u or v is zero and s(X,Y) has negative powers"); return SONIC_ERR_INEXACT_DIVISION; }
  const long NS = 2 * m + 2, K = 4 * m + 2;
  DevBuf S(sizeof(Fr) * NS), PR(sizeof(Fr) * 2 * NS), slots(sizeof(MsmSlot) * K), frout(sizeof(Fr) * (2 * m + 1)), flags(4), scaled;
  int* fl = flags.as<int>();
  HIP_OK(hipMemsetAsync(fl, 0, 4, st));
  HIP_OK(hipMemsetAsync(frout.p, 0, sizeof(Fr) * (2 * m + 1), st));
  BivTerms byx, byy;
  biv_upload(st, n_terms, x_exps, y_exps, coeffs, byx, fl);      // keeps X: s(X, y_j)
  biv_upload(st, n_terms, y_exps, x_exps, coeffs, byy, fl);      // keeps Y: s(u, Y)
  {
    std::vector<uint8_t> h(32 * (size_t)NS);
    for (long j = 0; j < m; j++) { memcpy(&h[32 * j], yzs + 64 * j, 32); memcpy(&h[32 * (m + j)], yzs + 64 * j + 32, 32); }
    memcpy(&h[32 * (2 * m)], u, 32); memcpy(&h[32 * (2 * m + 1)], v, 32);
    HIP_OK(hipMemcpyAsync(S.p, h.data(), h.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipStreamSynchronize(st));
  }
  fr_to_mont_enqueue(st, S.as<Fr>(), NS, fl);
  fr_with_inverse_enqueue(st, S.as<Fr>(), (int)NS, PR.as<Fr>());
  const Fr* P0 = PR.as<Fr>();
  auto pY = [&](long j) { return P0 + 2 * j; };
  auto pZ = [&](long j) { return P0 + 2 * (m + j); };
  const Fr *pU = P0 + 2 * (2 * m), *pV = P0 + 2 * (2 * m + 1);
  auto is_zero_pt = [&](const uint8_t* b) { return bytes_are_zero(b, 32); };
  MsmSlot* sl = slots.as<MsmSlot>();
  Fr* fo = frout.as<Fr>();
  DevBuf sy(sizeof(Fr) * byx.len), su(sizeof(Fr) * byy.len);
  Scratch sc[MSM_MAX_JOBS];
  MsmWorkspace& ws = lease.ws();
  // an opening at 0 is only defined without negative exponents (checked above), where it is a shift (open_job_at_zero)
  auto open_any = [&](Scratch& s_, const Fr* poly, long lo, long len, const Fr* zp, bool zero, Fr* fz, MsmSlot* slot) {
    return zero ? open_job_at_zero(st, srs, poly, len, fz ? fz : s_.fz_discard.as<Fr>(), slot, fl)
                : open_job(st, srs, s_, poly, lo, len, zp, fz, slot, fl);
  };
  for (auto& s_ : sc) s_.fz_discard.ensure(sizeof(Fr));
  // slots: S_j = 3j, W_j = 3j + 1, W'_j = 3j + 2;  Q_j = 3m + j;  Q_v = 4m;  C = 4m + 1.   frout: s_j = j, s'_j = m + j
  for (long j = 0; j < m; j++) {                                                       // Signature.hs:40-45, 54
    biv_eval_enqueue(st, byx, pY(j), scaled, sy.as<Fr>());                             // evalY y_j sXY
    MsmJob jobs[3];
    jobs[0] = commit_job(st, srs, sy.as<Fr>(), byx.lo, byx.len, d, &sl[3 * j], fl);
    jobs[1] = open_any(sc[1], sy.as<Fr>(), byx.lo, byx.len, pZ(j), is_zero_pt(yzs + 64 * j + 32), &fo[j], &sl[3 * j + 1]);
    jobs[2] = open_any(sc[2], sy.as<Fr>(), byx.lo, byx.len, pU, is_zero_pt(u), nullptr, &sl[3 * j + 2]);
    run_jobs(st, srs, ws, jobs, 3);
  }
  biv_eval_enqueue(st, byy, pU, scaled, su.as<Fr>());                                  // evalX u sXY          :51
  {
    MsmJob jobs[MSM_MAX_JOBS];
    int k = 0;
    auto flush = [&] { run_jobs(st, srs, ws, jobs, k); k = 0; };
    jobs[k++] = commit_job(st, srs, su.as<Fr>(), byy.lo, byy.len, d, &sl[4 * m + 1], fl);                              // C    :52
    for (long j = 0; j < m; j++) {                                                                                     // Q_j  :55
      if (k == MSM_MAX_JOBS) flush();
      jobs[k] = open_any(sc[k], su.as<Fr>(), byy.lo, byy.len, pY(j), is_zero_pt(yzs + 64 * j), &fo[m + j], &sl[3 * m + j]);
      k++;
    }
    if (k == MSM_MAX_JOBS) flush();
    jobs[k] = open_any(sc[k], su.as<Fr>(), byy.lo, byy.len, pV, is_zero_pt(v), &fo[2 * m], &sl[4 * m]);                // Q_v  :63
    k++;
    flush();
  }
  fr_from_mont_enqueue(st, fo, 2 * m + 1);
  std::vector<MsmSlot> hs((size_t)K);
  std::vector<uint8_t> hfr(32 * (size_t)(2 * m + 1));
  HIP_OK(hipMemcpyAsync(hs.data(), sl, sizeof(MsmSlot) * K, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(hfr.data(), fo, hfr.size(), hipMemcpyDeviceToHost, st));
  int hflags = read_flags(st, flags);
  if (hflags) return flags_to_status(hflags, "hscProve");
  std::vector<uint8_t> pts(96 * (size_t)K);
  {
    std::vector<G1XYZZ> sums((size_t)K);
    for (long i = 0; i < K; i++) sums[i] = msm_finish_host(hs[i]);
    g1_canonical_bytes_host_batch(sums.data(), (int)K, pts.data());
  }
  uint8_t* o = out;
  auto putG = [&](long i) { memcpy(o, &pts[96 * (size_t)i], 96); o += 96; };
  auto putF = [&](const uint8_t* b) { memcpy(o, b, 32); o += 32; };
  for (long j = 0; j < m; j++) { putG(3 * j); putF(&hfr[32 * j]); putG(3 * j + 1); }                 // hscS
  for (long j = 0; j < m; j++) { putF(&hfr[32 * (m + j)]); putG(3 * j + 2); putG(3 * m + j); }       // hscW
  putG(4 * m); putG(4 * m + 1); putF(u); putF(v);                                                   // Qv, C, u, v
  API_END
}

// ---- commitPoly / openPoly on caller-supplied sparse polynomials ------------------------------
struct DensePoly { DevBuf c; long lo = 0, len = 0; };

static int densify(hipStream_t st, const sonic_srs* srs, int64_t nt, const int64_t* exps, const uint8_t* coeffs, bool include_zero,
                   DensePoly& out, int* d_flags) {
  long lo = include_zero ? 0 : (nt ? exps[0] : 0), hi = lo;
  for (int64_t i = 0; i < nt; i++) { if (exps[i] < lo) lo = exps[i]; if (exps[i] > hi) hi = exps[i]; }
  const long d = srs_d(srs);
  if (hi - lo + 1 > 8 * (2 * d + 1) + 64) { set_error("polynomial exponent range [%ld, %ld] is far outside the SRS", lo, hi); return SONIC_ERR_SRS_INDEX; }
  out.lo = lo; out.len = hi - lo + 1;
  out.c.alloc(sizeof(Fr) * out.len);
  HIP_OK(hipMemsetAsync(out.c.p, 0, sizeof(Fr) * out.len, st));
  if (nt == 0) return SONIC_OK;
  std::vector<int64_t> order(nt);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return exps[a] < exps[b]; });
  std::vector<int64_t> se(nt);
  std::vector<uint8_t> sc(32 * (size_t)nt);
  for (int64_t i = 0; i < nt; i++) { se[i] = exps[order[i]]; memcpy(&sc[32 * (size_t)i], coeffs + 32 * order[i], 32); }
  DevBuf de(8 * (size_t)nt), dc(32 * (size_t)nt);
  HIP_OK(hipMemcpyAsync(de.p, se.data(), 8 * (size_t)nt, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(dc.p, sc.data(), 32 * (size_t)nt, hipMemcpyHostToDevice, st));
  fr_to_mont_enqueue(st, dc.as<Fr>(), nt, d_flags);
  sparse_to_dense_enqueue(st, de.as<int64_t>(), dc.as<Fr>(), nt, lo, out.c.as<Fr>());
  HIP_OK(hipStreamSynchronize(st));   // host staging vectors and de/dc go out of scope
  return SONIC_OK;
}

int sonic_commit_poly(const sonic_srs_t* srs, int64_t max, int64_t n_terms, const int64_t* exps, const uint8_t* coeffs, uint8_t out_g1[96]) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || n_terms < 0 || !out_g1 || (n_terms > 0 && (!exps || !coeffs))) return SONIC_ERR_INVALID_ARG;
  CallLease lease;
  hipStream_t st = lease.st();
  DevBuf flags(4), slot(sizeof(MsmSlot));
  HIP_OK(hipMemsetAsync(flags.p, 0, 4, st));
  DensePoly f;
  int rc = densify(st, srs, n_terms, exps, coeffs, false, f, flags.as<int>());
  if (rc) return rc;
  MsmJob job = commit_job(st, srs, f.c.as<Fr>(), f.lo, f.len, max, slot.as<MsmSlot>(), flags.as<int>());
  run_jobs(st, srs, lease.ws(), &job, 1);
  MsmSlot h;
  HIP_OK(hipMemcpyAsync(&h, slot.p, sizeof h, hipMemcpyDeviceToHost, st));
  int fl = read_flags(st, flags);
  if (fl) return flags_to_status(fl, "commitPoly");
  g1_canonical_bytes_host(msm_finish_host(h), out_g1);
  API_END
}

int sonic_open_poly(const sonic_srs_t* srs, const uint8_t z[32], int64_t n_terms, const int64_t* exps, const uint8_t* coeffs,
                    uint8_t out_fz[32], uint8_t out_g1[96]) {
  API_BEGIN_ON(srs_device(srs))
  if (!srs || !z || n_terms < 0 || !out_fz || !out_g1 || (n_terms > 0 && (!exps || !coeffs))) return SONIC_ERR_INVALID_ARG;
  CallLease lease;
  hipStream_t st = lease.st();
  DevBuf flags(4), slot(sizeof(MsmSlot)), zin(sizeof(Fr)), zpair(2 * sizeof(Fr)), fz(sizeof(Fr));
  HIP_OK(hipMemsetAsync(flags.p, 0, 4, st));
  DensePoly f;
  int rc = densify(st, srs, n_terms, exps, coeffs, true, f, flags.as<int>());
  if (rc) return rc;
  if (f.lo < 0 && bytes_are_zero(z, 32)) { set_error("openPoly: evaluation at z = 0 of a polynomial with negative exponents"); return SONIC_ERR_INEXACT_DIVISION; }
  HIP_OK(hipMemcpyAsync(zin.p, z, 32, hipMemcpyHostToDevice, st));
  fr_to_mont_enqueue(st, zin.as<Fr>(), 1, flags.as<int>());
  fr_with_inverse_enqueue(st, zin.as<Fr>(), 1, zpair.as<Fr>());
  Scratch sc;
  MsmJob job = bytes_are_zero(z, 32)
                   ? open_job_at_zero(st, srs, f.c.as<Fr>(), f.len, fz.as<Fr>(), slot.as<MsmSlot>(), flags.as<int>())
                   : open_job(st, srs, sc, f.c.as<Fr>(), f.lo, f.len, zpair.as<Fr>(), fz.as<Fr>(), slot.as<MsmSlot>(), flags.as<int>());
  run_jobs(st, srs, lease.ws(), &job, 1);
  fr_from_mont_enqueue(st, fz.as<Fr>(), 1);
  MsmSlot h;
  HIP_OK(hipMemcpyAsync(&h, slot.p, sizeof h, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(out_fz, fz.p, 32, hipMemcpyDeviceToHost, st));
  int fl = read_flags(st, flags);
  if (fl) return flags_to_status(fl, "openPoly");
  g1_canonical_bytes_host(msm_finish_host(h), out_g1);
  API_END
}

// ---- NTT / dense product -----------------------------------------------------------------------
// the twiddle tables of the stand-alone transforms: one set per device, used under that device's call mutex
static NttTables& shared_ntt() { DeviceCtx& c = current_ctx(); if (!c.ntt) c.ntt = new NttTables(); return *c.ntt; }

int sonic_ntt_fr(uint8_t* data, int log2n, int inverse) {
  API_BEGIN
  if (!data || log2n < 0 || log2n > 28) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  const long n = 1L << log2n;
  DevBuf d(sizeof(Fr) * n), flags(4);
  HIP_OK(hipMemsetAsync(flags.p, 0, 4, st));
  HIP_OK(hipMemcpyAsync(d.p, data, 32 * n, hipMemcpyHostToDevice, st));
  fr_to_mont_enqueue(st, d.as<Fr>(), n, flags.as<int>());
  if (log2n > 0) {
    shared_ntt().ensure(st, log2n);
    if (!inverse) { ntt_forward_enqueue(st, shared_ntt(), d.as<Fr>(), log2n); fr_bitrev_permute_enqueue(st, d.as<Fr>(), log2n); }
    else { fr_bitrev_permute_enqueue(st, d.as<Fr>(), log2n); ntt_inverse_enqueue(st, shared_ntt(), d.as<Fr>(), log2n); }
  }
  fr_from_mont_enqueue(st, d.as<Fr>(), n);
  int fl = read_flags(st, flags);
  if (fl) return flags_to_status(fl, "sonic_ntt_fr");
  HIP_OK(hipMemcpy(data, d.p, 32 * n, hipMemcpyDeviceToHost));
  API_END
}

int sonic_poly_mul_fr(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb, uint8_t* out) {
  API_BEGIN
  if (!a || !b || !out || na < 1 || nb < 1) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  const long rl = na + nb - 1;
  int lg = 0;
  while ((1L << lg) < rl) lg++;
  const long M = 1L << lg;
  DevBuf fa(sizeof(Fr) * M), fb(sizeof(Fr) * M), flags(4);
  HIP_OK(hipMemsetAsync(flags.p, 0, 4, st));
  HIP_OK(hipMemsetAsync(fa.p, 0, sizeof(Fr) * M, st));
  HIP_OK(hipMemsetAsync(fb.p, 0, sizeof(Fr) * M, st));
  HIP_OK(hipMemcpyAsync(fa.p, a, 32 * na, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(fb.p, b, 32 * nb, hipMemcpyHostToDevice, st));
  fr_to_mont_enqueue(st, fa.as<Fr>(), na, flags.as<int>());
  fr_to_mont_enqueue(st, fb.as<Fr>(), nb, flags.as<int>());
  if (lg > 0) {
    shared_ntt().ensure(st, lg);
    ntt_forward_enqueue(st, shared_ntt(), fa.as<Fr>(), lg);
    ntt_forward_enqueue(st, shared_ntt(), fb.as<Fr>(), lg);
  }
  if (lg > 0) ntt_inverse_of_product_enqueue(st, shared_ntt(), fa.as<Fr>(), fb.as<Fr>(), lg);
  else fr_pointwise_mul_enqueue(st, fa.as<Fr>(), fb.as<Fr>(), M);
  fr_from_mont_enqueue(st, fa.as<Fr>(), rl);
  int fl = read_flags(st, flags);
  if (fl) return flags_to_status(fl, "sonic_poly_mul_fr");
  HIP_OK(hipMemcpy(out, fa.p, 32 * rl, hipMemcpyDeviceToHost));
  API_END
}

// the same product with both operands and the result resident in HBM (canonical Fr, device pointers; d_out: na + nb - 1
// elements): what bench.py times for the NTT roofline -- the three transforms and the pointwise product alone on the chip
int sonic_poly_mul_fr_dev(const void* d_a, int64_t na, const void* d_b, int64_t nb, void* d_out) {
  API_BEGIN
  if (!d_a || !d_b || !d_out || na < 1 || nb < 1) return SONIC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(call_mutex());
  hipStream_t st = default_stream();
  const long rl = na + nb - 1;
  int lg = 0;
  while ((1L << lg) < rl) lg++;
  const long M = 1L << lg;
  DeviceCtx& dctx = current_ctx();
  DevBuf *fa = &dctx.mul_a, *fb = &dctx.mul_b, *flags = &dctx.mul_flags;       // the device's product scratch (under its call mutex)
  fa->ensure(sizeof(Fr) * M); fb->ensure(sizeof(Fr) * M); flags->ensure(4);
  HIP_OK(hipMemsetAsync(flags->p, 0, 4, st));
  HIP_OK(hipMemsetAsync(fa->p, 0, sizeof(Fr) * M, st));
  HIP_OK(hipMemsetAsync(fb->p, 0, sizeof(Fr) * M, st));
  HIP_OK(hipMemcpyAsync(fa->p, d_a, 32 * na, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(fb->p, d_b, 32 * nb, hipMemcpyDeviceToDevice, st));
  fr_to_mont_enqueue(st, fa->as<Fr>(), na, flags->as<int>());
  fr_to_mont_enqueue(st, fb->as<Fr>(), nb, flags->as<int>());
  if (lg > 0) {
    shared_ntt().ensure(st, lg);
    ntt_forward_enqueue(st, shared_ntt(), fa->as<Fr>(), lg);
    ntt_forward_enqueue(st, shared_ntt(), fb->as<Fr>(), lg);
  }
  if (lg > 0) ntt_inverse_of_product_enqueue(st, shared_ntt(), fa->as<Fr>(), fb->as<Fr>(), lg);
  else fr_pointwise_mul_enqueue(st, fa->as<Fr>(), fb->as<Fr>(), M);
  fr_from_mont_enqueue(st, fa->as<Fr>(), rl);
  HIP_OK(hipMemcpyAsync(d_out, fa->p, 32 * rl, hipMemcpyDeviceToDevice, st));
  int fl = read_flags(st, *flags);
  if (fl) return flags_to_status(fl, "sonic_poly_mul_fr_dev");
  API_END
}

}  // extern "C"
