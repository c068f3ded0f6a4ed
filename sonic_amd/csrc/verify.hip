// The verifier side of the reference's API, on the host CPU: pcV (src/Sonic/CommitmentScheme.hs:51-68),
// hscVerify (src/Sonic/Signature.hs:74-90) and verify (src/Sonic/Protocol.hs:111-130).
//
// This is outside the accelerated hot path: (4+3Q) x 3 pairings and O(nQ) field work per
// proof, independent of the MSM sizes, so it runs on the host with the same limb headers the kernels use.
// The three G2 elements it needs come from the GPU-generated SRS (sonic_srs_get_g2_points).
//
// Pairing: only equalities of pairing products are tested (eA <> eB == eC), so any bilinear non-degenerate
// pairing on (G1, G2) accepts exactly the same proofs as pairing-1.0.0's.  Used here (pairing.hpp): the ate Miller loop over
// |x| = 0xd201000000010000 with the running point on the twist and sparse lines in Fq12 = Fq2[w]/(w^6 - (1+u)) as a
// 2-3-2 tower, one shared final exponentiation 3 (q^12 - 1)/r per check: ~6 ms per pcV on one host core (the plain
// polynomial-basis pairing of rounds 1-2, now tests/pairing_plain.hpp, took 170 ms).
#include <string.h>
#include <algorithm>
#include <system_error>
#include <thread>
#include <vector>
#include "internal.hpp"
#include "g2.hpp"
#include "pairing.hpp"
#include "fs.hpp"

namespace sonic {
namespace {

// ---- host-side group / field helpers -------------------------------------------------------------------
G1XYZZ g1_mul_fr(const G1Affine& p, const Fr& k_std) {
  G1XYZZ acc = G1XYZZ::inf();
  const G1XYZZ base = G1XYZZ::from_affine(p);
  bool started = false;
  for (int i = 255; i >= 0; i--) {
    if (started) acc = g1_dbl(acc);
    if ((k_std.l[i >> 5] >> (i & 31)) & 1) { acc = g1_add(acc, base); started = true; }
  }
  return acc;
}
G1Affine g1_gen_host() {
  constexpr uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  G1Affine g;
  for (int i = 0; i < 12; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
  return g;
}
bool load_fr(const uint8_t* b, Fr& mont) { Fr s; memcpy(s.l, b, 32); if (!fp_is_canonical(s)) return false; mont = fp_to_mont(s); return true; }
bool load_g1(const uint8_t* b, G1Affine& p) {
  memcpy(p.x.l, b, 48); memcpy(p.y.l, b + 48, 48);
  if (p.is_inf()) return true;
  if (!fp_is_canonical(p.x) || !fp_is_canonical(p.y)) return false;
  p.x = fp_to_mont(p.x); p.y = fp_to_mont(p.y);
  Fq four = fp_dbl(fp_dbl(Fq::one()));
  if (!(fp_sqr(p.y) == fp_add(fp_mul(fp_sqr(p.x), p.x), four))) return false;
  // E(Fq) has cofactor points (e.g. (0, 2), order 3) and the pairing is bilinear only on the order-r subgroup: points that
  // come from a prover must satisfy r P = O before they reach the Miller loop
  constexpr uint32_t rl[8] = FR_P;
  Fr r_std; for (int i = 0; i < 8; i++) r_std.l[i] = rl[i];
  return g1_mul_fr(p, r_std).is_inf();
}
bool load_g2(const uint8_t* b, G2Affine& p) {
  memcpy(p.x.c0.l, b, 48); memcpy(p.x.c1.l, b + 48, 48); memcpy(p.y.c0.l, b + 96, 48); memcpy(p.y.c1.l, b + 144, 48);
  if (p.is_inf()) return true;
  p.x.c0 = fp_to_mont(p.x.c0); p.x.c1 = fp_to_mont(p.x.c1); p.y.c0 = fp_to_mont(p.y.c0); p.y.c1 = fp_to_mont(p.y.c1);
  return true;
}
Fr fr_pow(const Fr& a, uint64_t e) { return fp_pow_u64(a, e); }

struct VerifierKey { G2Affine h_alpha, h_alpha_x; };   // hPositiveAlphaX[0], [1]

int fetch_g2(const sonic_srs* srs, int basis, int64_t e, G2Affine& out) {
  uint8_t b[192];
  int rc = sonic_srs_get_g2_points(srs, basis, e, 1, b);
  if (rc) return rc;
  load_g2(b, out);
  // fail closed: the Miller loop of a G2 element at infinity is 1, so a verifier key at infinity would accept anything.  No valid
  // SRS (x, alpha != 0) holds one; sonic_srs_set_g2_points / sonic_srs_load refuse them, this is the second line of defence.
  if (out.is_inf()) { set_error("verifier: the SRS holds the point at infinity as G2 element (basis %d, exponent %ld)", basis, (long)e); return SONIC_ERR_BAD_ENCODING; }
  return SONIC_OK;
}

// pcV srs max F z (v, W)  (CommitmentScheme.hs:51-68), in two steps: the G2 element h^{x^{-d+max}} comes from the SRS handle
// (device memory, the library's call mutex), the pairing equation itself is pure host arithmetic -- so a verifier's checks
// fetch their elements first and then run side by side on host threads (a proof with Q constraints has 4 + 3Q of them at
// ~6-9 ms each: three Miller loops, one final exponentiation, two scalar multiples in G1).
int pc_v_element(const sonic_srs* srs, int64_t maxm, G2Affine& hxi) {
  const int64_t d = srs_d(srs);
  const int64_t difference = -d + maxm;                              // h^{x^{-d+max}}: hPositiveX / hNegativeX
  if (difference > d || difference < -d) { set_error("pcV: hPositiveX / hNegativeX is not long enough: %ld", (long)difference); return SONIC_ERR_SRS_INDEX; }
  return fetch_g2(srs, 0, difference, hxi);
}
bool pc_v_equation(const VerifierKey& vk, const G2Affine& hxi, const G1Affine& F, const Fr& z_m, const Fr& v_m, const G1Affine& W) {
  const Fr v = fp_from_mont(v_m), negz = fp_from_mont(fp_neg(z_m));
  G1Affine left = g1_to_affine(g1_add(g1_mul_fr(g1_gen_host(), v), g1_mul_fr(W, negz)));   // g^v W^{-z}
  G1Affine negF = g1_neg(F);
  using namespace pairing;
  const F12 f = f12_mul(f12_mul(miller_loop(W, vk.h_alpha_x), miller_loop(left, vk.h_alpha)), miller_loop(negF, hxi));
  return final_exponentiation(f).is_one();                           // eA <> eB == eC
}
int pc_v(const sonic_srs* srs, const VerifierKey& vk, int64_t maxm, const G1Affine& F, const Fr& z_m, const Fr& v_m, const G1Affine& W, bool& ok) {
  G2Affine hxi;
  int rc = pc_v_element(srs, maxm, hxi);
  if (rc) return rc;
  ok = pc_v_equation(vk, hxi, F, z_m, v_m, W);
  return SONIC_OK;
}

struct PcvCheck { int64_t maxm; G1Affine F; Fr z, val; G1Affine W; };
// all checks of a verifier: elements first (one per distinct max), equations on up to 16 host threads
int run_checks(const sonic_srs* srs, const VerifierKey& vk, const std::vector<PcvCheck>& checks, bool& all) {
  std::vector<int64_t> maxs;
  std::vector<G2Affine> elems;
  std::vector<int> which(checks.size());
  for (size_t i = 0; i < checks.size(); i++) {
    size_t k = 0;
    while (k < maxs.size() && maxs[k] != checks[i].maxm) k++;
    if (k == maxs.size()) {
      G2Affine h;
      int rc = pc_v_element(srs, checks[i].maxm, h);
      if (rc) return rc;
      maxs.push_back(checks[i].maxm); elems.push_back(h);
    }
    which[i] = (int)k;
  }
  std::vector<char> ok(checks.size(), 0);
  const int nt = (int)std::min<size_t>(checks.size(), 16);
  auto work = [&](int w, int stride) {
    for (size_t i = w; i < checks.size(); i += stride)
      ok[i] = pc_v_equation(vk, elems[which[i]], checks[i].F, checks[i].z, checks[i].val, checks[i].W) ? 1 : 0;
  };
  ThreadGroup th;
  int started = 0;
  try {
    for (; started < nt; started++) th.emplace_back(work, started, nt);
  } catch (const std::system_error&) {}        // thread limit of the host process: the calling thread takes the rest
  for (int w = started; w < nt; w++) work(w, nt);
  for (auto& t : th) t.join();
  for (char c : ok) all = all && c;
  return SONIC_OK;
}

// the 3m + 1 pcV checks of hscVerify (Signature.hs:82-89) once s(u,v) is known
void hsc_push_checks(int64_t d, int64_t m, const std::vector<Fr>& ys, const std::vector<Fr>& zs, const std::vector<G1Affine>& Sj, const std::vector<Fr>& sj,
                     const std::vector<G1Affine>& Wj, const std::vector<Fr>& spj, const std::vector<G1Affine>& Wpj, const std::vector<G1Affine>& Qj,
                     const G1Affine& Qv, const G1Affine& C, const Fr& u, const Fr& v, const Fr& sv, std::vector<PcvCheck>& checks) {
  for (int64_t j = 0; j < m; j++) {                                // Signature.hs:82-88
    checks.push_back(PcvCheck{d, Sj[j], zs[j], sj[j], Wj[j]});
    checks.push_back(PcvCheck{d, Sj[j], u, spj[j], Wpj[j]});
    checks.push_back(PcvCheck{d, C, ys[j], spj[j], Qj[j]});
  }
  checks.push_back(PcvCheck{d, C, v, sv, Qv});                     // Signature.hs:89
}

// hscVerify srs sXY yzs proof (Signature.hs:74-90) for the s(X,Y) of a circuit (Constraints.hs:34-53): s(u,v) on the host,
// then its 3m + 1 pcV checks are appended to `checks` (run_checks evaluates them).
int hsc_checks(const sonic_srs* srs, const VerifierKey& vk, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
               int64_t m, const std::vector<Fr>& ys, const std::vector<Fr>& zs, const std::vector<G1Affine>& Sj, const std::vector<Fr>& sj,
               const std::vector<G1Affine>& Wj, const std::vector<Fr>& spj, const std::vector<G1Affine>& Wpj, const std::vector<G1Affine>& Qj,
               const G1Affine& Qv, const G1Affine& C, const Fr& u, const Fr& v, std::vector<PcvCheck>& checks) {
  // s(u, v): sum_i u^-i U_i(v) + u^i V_i(v) + u^{i+n} W_i(v)   (Signature.hs:81; Constraints.hs:34-53)
  if (u.is_zero() || v.is_zero()) { set_error("hscVerify: u or v is zero"); return SONIC_ERR_INEXACT_DIVISION; }
  std::vector<Fr> vq(Q);
  { Fr x = fr_pow(v, (uint64_t)n); for (int64_t q = 0; q < Q; q++) { x = fp_mul(x, v); vq[q] = x; } }
  const Fr uinv = fp_inv(u), vinv = fp_inv(v), un = fr_pow(u, (uint64_t)n);
  Fr up = Fr::one(), um = Fr::one(), vp = Fr::one(), vm = Fr::one(), sv = Fr::zero();
  for (int64_t i = 1; i <= n; i++) {
    up = fp_mul(up, u); um = fp_mul(um, uinv); vp = fp_mul(vp, v); vm = fp_mul(vm, vinv);
    Fr Ui = Fr::zero(), Vi = Fr::zero(), Wi = Fr::zero(), c;
    for (int64_t q = 0; q < Q; q++) {
      if (!load_fr(wL + 32 * (q * n + i - 1), c)) return SONIC_ERR_BAD_ENCODING; Ui = fp_add(Ui, fp_mul(c, vq[q]));
      if (!load_fr(wR + 32 * (q * n + i - 1), c)) return SONIC_ERR_BAD_ENCODING; Vi = fp_add(Vi, fp_mul(c, vq[q]));
      if (!load_fr(wO + 32 * (q * n + i - 1), c)) return SONIC_ERR_BAD_ENCODING; Wi = fp_add(Wi, fp_mul(c, vq[q]));
    }
    Wi = fp_sub(fp_sub(Wi, vp), vm);
    sv = fp_add(sv, fp_add(fp_add(fp_mul(um, Ui), fp_mul(up, Vi)), fp_mul(fp_mul(up, un), Wi)));
  }
  hsc_push_checks(srs_d(srs), m, ys, zs, Sj, sj, Wj, spj, Wpj, Qj, Qv, C, u, v, sv, checks);
  return SONIC_OK;
}

}  // namespace
}  // namespace sonic

using namespace sonic;

extern "C" {

int sonic_pc_v(const sonic_srs_t* srs, int64_t max, const uint8_t commitment[96], const uint8_t z[32], const uint8_t v[32],
               const uint8_t w[96], int* accepted) {
  try {
    if (!srs || !commitment || !z || !v || !w || !accepted) return SONIC_ERR_INVALID_ARG;
    G1Affine F, W; Fr zm, vm;
    if (!load_g1(commitment, F) || !load_g1(w, W) || !load_fr(z, zm) || !load_fr(v, vm)) { set_error("pcV: bad encoding"); return SONIC_ERR_BAD_ENCODING; }
    VerifierKey vk;
    int rc = fetch_g2(srs, 1, 0, vk.h_alpha);
    if (!rc) rc = fetch_g2(srs, 1, 1, vk.h_alpha_x);
    if (rc) return rc;
    bool ok = false;
    rc = pc_v(srs, vk, max, F, zm, vm, W, ok);
    *accepted = ok ? 1 : 0;
    return rc;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}

// verify srs circuit proof y z yzs  (Protocol.hs:111-130); yzs = Q pairs (y_j, z_j), 64 bytes each
int sonic_verify(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                 const uint8_t* cs, const uint8_t* proof, const uint8_t y[32], const uint8_t z[32], const uint8_t* yzs, int* accepted) {
  try {
    if (!srs || n < 1 || Q < 1 || !wL || !wR || !wO || !cs || !proof || !y || !z || !yzs || !accepted) return SONIC_ERR_INVALID_ARG;
    *accepted = 0;
    const uint8_t* p = proof;
    auto G = [&](G1Affine& o) { bool k = load_g1(p, o); p += 96; return k; };
    auto F = [&](Fr& o) { bool k = load_fr(p, o); p += 32; return k; };
    G1Affine R, T, Wa, Wb, Wt, Qv, C;
    Fr a, b, s, u, v, ym, zm;
    bool enc = G(R) && G(T) && F(a) && G(Wa) && F(b) && G(Wb) && G(Wt) && F(s);
    std::vector<G1Affine> Sj(Q), Wj(Q), Wpj(Q), Qj(Q);
    std::vector<Fr> sj(Q), spj(Q), ys(Q), zs(Q);
    for (int64_t j = 0; j < Q; j++) enc = enc && G(Sj[j]) && F(sj[j]) && G(Wj[j]);
    for (int64_t j = 0; j < Q; j++) enc = enc && F(spj[j]) && G(Wpj[j]) && G(Qj[j]);
    enc = enc && G(Qv) && G(C) && F(u) && F(v) && load_fr(y, ym) && load_fr(z, zm);
    for (int64_t j = 0; j < Q; j++) enc = enc && load_fr(yzs + 64 * j, ys[j]) && load_fr(yzs + 64 * j + 32, zs[j]);
    if (!enc) { set_error("verify: non-canonical field element, or point off the curve or outside the order-r subgroup"); return SONIC_ERR_BAD_ENCODING; }
    // k(y) = sum_q cs[q] y^{n+q}                                  (Constraints.hs:67-68)
    Fr ky = Fr::zero(), pw = fr_pow(ym, (uint64_t)n);
    for (int64_t q = 0; q < Q; q++) { Fr c; if (!load_fr(cs + 32 * q, c)) return SONIC_ERR_BAD_ENCODING; pw = fp_mul(pw, ym); ky = fp_add(ky, fp_mul(c, pw)); }
    const Fr t = fp_sub(fp_mul(a, fp_add(b, s)), ky);              // Protocol.hs:120
    VerifierKey vk;
    int rc = fetch_g2(srs, 1, 0, vk.h_alpha);
    if (!rc) rc = fetch_g2(srs, 1, 1, vk.h_alpha_x);
    if (rc) return rc;
    const int64_t d = srs_d(srs);
    std::vector<PcvCheck> checks;
    rc = hsc_checks(srs, vk, n, Q, wL, wR, wO, Q, ys, zs, Sj, sj, Wj, spj, Wpj, Qj, Qv, C, u, v, checks);   // hscVerify, Signature.hs:74-90
    if (rc) return rc;
    checks.push_back(PcvCheck{n, R, zm, a, Wa});                   // Protocol.hs:123
    checks.push_back(PcvCheck{n, R, fp_mul(ym, zm), b, Wb});       // :124
    checks.push_back(PcvCheck{d, T, zm, t, Wt});                   // :125
    bool all = true;
    rc = run_checks(srs, vk, checks, all);
    if (rc) return rc;
    *accepted = all ? 1 : 0;
    return SONIC_OK;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}

// what ties a Fiat-Shamir transcript to ONE reference string (fs.hpp): four G1 elements that determine x and alpha
// (constant for a handle: computed on first use -- four point fetches from the device -- and cached in the handle, srs_cached_id)
static int make_srs_id(const sonic_srs* srs, uint8_t out[32]) {
  try {
    uint8_t pts[4 * 96];
    int rc = sonic_srs_get_points(srs, 0, 1, 1, pts);                 // g^x            gPositiveX[1]
    if (!rc) rc = sonic_srs_get_points(srs, 1, 1, 1, pts + 96);       // g^{alpha x}    gPositiveAlphaX[0]
    if (!rc) rc = sonic_srs_get_points(srs, 0, -1, 1, pts + 192);     // g^{1/x}        gNegativeX[0]
    if (!rc) rc = sonic_srs_get_points(srs, 1, -1, 1, pts + 288);     // g^{alpha/x}    gNegativeAlphaX[0]
    if (rc) return rc;
    fs_srs_id_of_points(srs_d(srs), pts, out);
    return SONIC_OK;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}
int sonic_fs_srs_id(const sonic_srs_t* srs, uint8_t out[32]) {
  if (!srs || !out) return SONIC_ERR_INVALID_ARG;
  return srs_cached_id(srs, &make_srs_id, out);
}

// srsPairing = pairing gen (hPositiveAlphaX !! 0) = e(g, h^alpha) (SRS.hs:21,42): the reduced ate pairing
// f_{x, Q}(P)^((q^12 - 1)/r) with the (negative) curve parameter x.  The verifier's own pairing (pairing.hpp) computes
// v = f_{|x|, Q}(P)^(3 (q^12 - 1)/r) -- any non-degenerate bilinear map serves its equality tests -- so the record field is derived
// from it exactly: v lies in the order-r subgroup, v^(1/3 mod r) = f_{|x|,Q}(P)^((q^12-1)/r), and the sign of x turns that into its
// inverse, which in the cyclotomic subgroup is the conjugate.  (Which representative pairing-1.0.0 itself returns is [dep, unverified]:
// the package is not in the reference tree; this is the textbook definition.)
int sonic_srs_pairing(const sonic_srs_t* srs, uint8_t out[576]) {
  try {
    if (!srs || !out) return SONIC_ERR_INVALID_ARG;
    G2Affine ha;
    int rc = fetch_g2(srs, 1, 0, ha);                                  // hPositiveAlphaX[0] = h^alpha
    if (rc) return rc;
    using namespace pairing;
    const F12 v = final_exponentiation(miller_loop(g1_gen_host(), ha));
    // 1/3 mod r as an integer: Fr arithmetic of the limb headers
    Fr three = fp_add(fp_add(Fr::one(), Fr::one()), Fr::one());
    const Fr e = fp_from_mont(fp_inv(three));
    F12 acc = F12::one();
    for (int i = 255; i >= 0; i--) {
      acc = f12_sqr(acc);
      if ((e.l[i >> 5] >> (i & 31)) & 1) acc = f12_mul(acc, v);
    }
    const F12 res = f12_conj(acc);
    const F6* halves[2] = {&res.c0, &res.c1};
    uint8_t* o = out;
    for (int i = 0; i < 2; i++) {
      const Fq2* cs[3] = {&halves[i]->a0, &halves[i]->a1, &halves[i]->a2};
      for (int j = 0; j < 3; j++) {
        const Fq c0 = fp_from_mont(cs[j]->c0), c1 = fp_from_mont(cs[j]->c1);
        memcpy(o, c0.l, 48); memcpy(o + 48, c1.l, 48); o += 96;
      }
    }
    return SONIC_OK;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}

// verify for a proof made by sonic_prover_prove_fs: the challenges y, z, (y_j, z_j) are not handed over (RndOracle) but recomputed from
// the statement and the proof (fs.hpp), and the proof's u, v must be the ones its own transcript yields
int sonic_verify_fs(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                    const uint8_t* cs, const uint8_t* proof, int* accepted) {
  try {
    if (!srs || n < 1 || Q < 1 || !wL || !wR || !wO || !cs || !proof || !accepted) return SONIC_ERR_INVALID_ARG;
    *accepted = 0;
    uint8_t digest[32];
    int rc = sonic_fs_circuit_digest(n, Q, wL, wR, wO, cs, digest);
    if (rc) return rc;
    uint8_t srs_id[32];
    rc = sonic_fs_srs_id(srs, srs_id);
    if (rc) return rc;
    std::vector<uint8_t> ch(32 * (size_t)(4 + 2 * Q));
    fs_challenges_of_proof(n, Q, srs_d(srs), digest, srs_id, proof, ch.data());
    const uint8_t* uv = proof + sonic_proof_size(Q) - 64;
    if (memcmp(uv, &ch[32 * (2 + 2 * Q)], 64) != 0) return SONIC_OK;          // u, v are not this transcript's: rejected
    std::vector<uint8_t> yzs(64 * (size_t)Q);
    for (int64_t j = 0; j < Q; j++) { memcpy(&yzs[64 * j], &ch[32 * (2 + j)], 32); memcpy(&yzs[64 * j + 32], &ch[32 * (2 + Q + j)], 32); }
    return sonic_verify(srs, n, Q, wL, wR, wO, cs, proof, &ch[0], &ch[32], yzs.data(), accepted);
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}

// hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool (Signature.hs:74-90) for the s(X,Y) of a circuit;
// hsc = the bytes sonic_prover_hsc_prove wrote (m pairs)
int sonic_hsc_verify(const sonic_srs_t* srs, int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO,
                     int64_t m, const uint8_t* yzs, const uint8_t* hsc, int* accepted) {
  try {
    if (!srs || n < 1 || Q < 1 || !wL || !wR || !wO || m < 0 || (m > 0 && !yzs) || !hsc || !accepted) return SONIC_ERR_INVALID_ARG;
    *accepted = 0;
    const uint8_t* p = hsc;
    auto G = [&](G1Affine& o) { bool k = load_g1(p, o); p += 96; return k; };
    auto F = [&](Fr& o) { bool k = load_fr(p, o); p += 32; return k; };
    std::vector<G1Affine> Sj(m), Wj(m), Wpj(m), Qj(m);
    std::vector<Fr> sj(m), spj(m), ys(m), zs(m);
    G1Affine Qv, C; Fr u, v;
    bool enc = true;
    for (int64_t j = 0; j < m; j++) enc = enc && G(Sj[j]) && F(sj[j]) && G(Wj[j]);
    for (int64_t j = 0; j < m; j++) enc = enc && F(spj[j]) && G(Wpj[j]) && G(Qj[j]);
    enc = enc && G(Qv) && G(C) && F(u) && F(v);
    for (int64_t j = 0; j < m; j++) enc = enc && load_fr(yzs + 64 * j, ys[j]) && load_fr(yzs + 64 * j + 32, zs[j]);
    if (!enc) { set_error("hscVerify: non-canonical field element, or point off the curve or outside the order-r subgroup"); return SONIC_ERR_BAD_ENCODING; }
    VerifierKey vk;
    int rc = fetch_g2(srs, 1, 0, vk.h_alpha);
    if (!rc) rc = fetch_g2(srs, 1, 1, vk.h_alpha_x);
    if (rc) return rc;
    bool all = true;
    std::vector<PcvCheck> checks;
    rc = hsc_checks(srs, vk, n, Q, wL, wR, wO, m, ys, zs, Sj, sj, Wj, spj, Wpj, Qj, Qv, C, u, v, checks);
    if (!rc) rc = run_checks(srs, vk, checks, all);
    if (rc) return rc;
    *accepted = all ? 1 : 0;
    return SONIC_OK;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}


// hscVerify :: SRS -> BiVLaurent Fr -> [(Fr, Fr)] -> HscProof -> Bool (Signature.hs:74-90) for any sparse bivariate Laurent polynomial
// (the counterpart of sonic_hsc_prove_poly): s(u,v) = eval (evalY v sXY) u on the host, then the 3m + 1 pcV checks
int sonic_hsc_verify_poly(const sonic_srs_t* srs, int64_t n_terms, const int64_t* x_exps, const int64_t* y_exps, const uint8_t* coeffs,
                          int64_t m, const uint8_t* yzs, const uint8_t* hsc, int* accepted) {
  try {
    if (!srs || n_terms < 0 || (n_terms > 0 && (!x_exps || !y_exps || !coeffs)) || m < 0 || (m > 0 && !yzs) || !hsc || !accepted) return SONIC_ERR_INVALID_ARG;
    *accepted = 0;
    const uint8_t* p = hsc;
    auto G = [&](G1Affine& o) { bool k = load_g1(p, o); p += 96; return k; };
    auto F = [&](Fr& o) { bool k = load_fr(p, o); p += 32; return k; };
    std::vector<G1Affine> Sj(m), Wj(m), Wpj(m), Qj(m);
    std::vector<Fr> sj(m), spj(m), ys(m), zs(m);
    G1Affine Qv, C; Fr u, v;
    bool enc = true;
    for (int64_t j = 0; j < m; j++) enc = enc && G(Sj[j]) && F(sj[j]) && G(Wj[j]);
    for (int64_t j = 0; j < m; j++) enc = enc && F(spj[j]) && G(Wpj[j]) && G(Qj[j]);
    enc = enc && G(Qv) && G(C) && F(u) && F(v);
    for (int64_t j = 0; j < m; j++) enc = enc && load_fr(yzs + 64 * j, ys[j]) && load_fr(yzs + 64 * j + 32, zs[j]);
    if (!enc) { set_error("hscVerify: non-canonical field element, or point off the curve or outside the order-r subgroup"); return SONIC_ERR_BAD_ENCODING; }
    Fr sv = Fr::zero();
    const Fr uinv = u.is_zero() ? u : fp_inv(u), vinv = v.is_zero() ? v : fp_inv(v);
    for (int64_t i = 0; i < n_terms; i++) {
      Fr c;
      if (!load_fr(coeffs + 32 * i, c)) { set_error("hscVerify: non-canonical coefficient"); return SONIC_ERR_BAD_ENCODING; }
      const int64_t ex = x_exps[i], ey = y_exps[i];
      if ((ex < 0 && u.is_zero()) || (ey < 0 && v.is_zero())) { set_error("hscVerify: u or v is zero and s(X,Y) has negative powers"); return SONIC_ERR_INEXACT_DIVISION; }
      const Fr px = fr_pow(ex >= 0 ? u : uinv, (uint64_t)(ex >= 0 ? ex : -ex)), py = fr_pow(ey >= 0 ? v : vinv, (uint64_t)(ey >= 0 ? ey : -ey));
      sv = fp_add(sv, fp_mul(c, fp_mul(px, py)));
    }
    VerifierKey vk;
    int rc = fetch_g2(srs, 1, 0, vk.h_alpha);
    if (!rc) rc = fetch_g2(srs, 1, 1, vk.h_alpha_x);
    if (rc) return rc;
    bool all = true;
    std::vector<PcvCheck> checks;
    hsc_push_checks(srs_d(srs), m, ys, zs, Sj, sj, Wj, spj, Wpj, Qj, Qv, C, u, v, sv, checks);
    rc = run_checks(srs, vk, checks, all);
    if (rc) return rc;
    *accepted = all ? 1 : 0;
    return SONIC_OK;
  } catch (const HipFail& f) { return f.code; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SONIC_ERR_HIP; }
}

}  // extern "C"
