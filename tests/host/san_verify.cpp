// AddressSanitizer / UBSan driver for the verifier's host path: sonic_amd/csrc/verify.hip compiled as plain C++ by g++ (it holds no
// kernel: pcV, hscVerify and verify -- src/Sonic/CommitmentScheme.hs:51-68, Signature.hs:74-90, Protocol.hs:111-130 -- are host
// code over pairing.hpp), linked against stand-ins for the three things it takes from the device side: the SRS handle's degree, its
// G2 elements (computed here from a known trapdoor with the same host group code) and the error slot.
//   san_verify <case file>      case file (written by tests/test_sanitizers.py from tests/golden/prove_small.json):
//       i64 n, Q, d | x, alpha (32 B each) | wL, wR, wO (Q n x 32 B each) | cs (Q x 32) | proof | y, z (32 each) | yzs (Q x 64) |
//       srsPairing = e(g, h^alpha) as oracle/pairing.py computes it, in sonic_srs_pairing's layout (576 B)
// Runs sonic_verify on the golden proof (must accept), on proofs with one byte changed in every field (must reject or report a bad
// encoding), with other challenges, sonic_pc_v on a hand-made opening, and sonic_verify_fs.  Prints "san_verify ok".
#include <stdarg.h>
#include <stdio.h>
#include <map>
#include <vector>
#include "../../sonic_amd/csrc/verify.hip"

struct sonic_srs { int64_t d; sonic::Fr x, alpha; std::map<std::pair<int, int64_t>, sonic::G2Affine> cache; };

namespace sonic {
static char g_msg[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_msg, sizeof g_msg, fmt, ap); va_end(ap); }
int64_t srs_d(const sonic_srs* s) { return s->d; }
int srs_cached_id(const sonic_srs* s, int (*make)(const sonic_srs*, uint8_t*), uint8_t out[32]) { return make(s, out); }      // (the handle's cache lives in api.hip)
}  // namespace sonic
using namespace sonic;

static G2Affine g2_gen_host() {
  constexpr uint32_t x0[12] = G2_GEN_X0_MONT, x1[12] = G2_GEN_X1_MONT, y0[12] = G2_GEN_Y0_MONT, y1[12] = G2_GEN_Y1_MONT;
  G2Affine g;
  for (int i = 0; i < 12; i++) { g.x.c0.l[i] = x0[i]; g.x.c1.l[i] = x1[i]; g.y.c0.l[i] = y0[i]; g.y.c1.l[i] = y1[i]; }
  return g;
}
static G2Affine g2_mul_fr(const G2Affine& p, const Fr& k_std) {
  G2Jac acc = G2Jac::inf();
  for (int i = 254; i >= 0; i--) { acc = g2_dbl(acc); if ((k_std.l[i >> 5] >> (i & 31)) & 1) acc = g2_add_mixed(acc, p); }
  return g2_to_affine(acc);
}
static Fr pow_signed(const Fr& a_mont, int64_t e) { return fp_pow_u64(e >= 0 ? a_mont : fp_inv(a_mont), (uint64_t)(e >= 0 ? e : -e)); }

extern "C" {
size_t sonic_proof_size(int64_t Q) { return (size_t)((7 + 4 * Q) * 96 + (5 + 2 * Q) * 32); }
int sonic_fs_circuit_digest(int64_t n, int64_t Q, const uint8_t* wL, const uint8_t* wR, const uint8_t* wO, const uint8_t* cs, uint8_t out[32]) {
  Sha256 h;
  h.update("sonic-hip/circuit/v1", 20);
  FsTranscript::le64(h, n); FsTranscript::le64(h, Q);
  h.update(wL, (size_t)(32 * Q * n)); h.update(wR, (size_t)(32 * Q * n)); h.update(wO, (size_t)(32 * Q * n)); h.update(cs, (size_t)(32 * Q));
  h.finish(out);
  return 0;
}
// g^{alpha^basis x^e} from the trapdoor, canonical bytes (the Fiat-Shamir verifier reads four of them for the srs id, fs.hpp)
int sonic_srs_get_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out) {
  constexpr uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  G1Affine g;
  for (int i = 0; i < 12; i++) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
  for (int64_t i = 0; i < n; i++) {
    const int64_t e = e0 + i;
    if (e < -srs->d || e > srs->d) return SONIC_ERR_SRS_INDEX;
    if (basis && e == 0) { memset(out + 96 * i, 0, 96); continue; }
    Fr k = pow_signed(srs->x, e);
    if (basis) k = fp_mul(k, srs->alpha);
    const Fr ks = fp_from_mont(k);
    G1XYZZ acc = G1XYZZ::inf();
    for (int b = 254; b >= 0; b--) { acc = g1_dbl(acc); if ((ks.l[b >> 5] >> (b & 31)) & 1) acc = g1_add_mixed(acc, g); }
    g1_canonical_bytes_host(acc, out + 96 * i);
  }
  return 0;
}
// h^{alpha^basis x^e} from the trapdoor, canonical bytes (what the device-side SRS serves)
int sonic_srs_get_g2_points(const sonic_srs_t* srs, int basis, int64_t e0, int64_t n, uint8_t* out) {
  sonic_srs* s = const_cast<sonic_srs*>(srs);
  for (int64_t i = 0; i < n; i++) {
    const int64_t e = e0 + i;
    if (e < -s->d || e > s->d) return SONIC_ERR_SRS_INDEX;
    auto key = std::make_pair(basis, e);
    if (!s->cache.count(key)) {
      Fr k = pow_signed(s->x, e);
      if (basis) k = fp_mul(k, s->alpha);
      s->cache[key] = g2_mul_fr(g2_gen_host(), fp_from_mont(k));
    }
    const G2Affine p = s->cache[key];
    const Fq c[4] = {fp_from_mont(p.x.c0), fp_from_mont(p.x.c1), fp_from_mont(p.y.c0), fp_from_mont(p.y.c1)};
    for (int j = 0; j < 4; j++) memcpy(out + 192 * i + 48 * j, c[j].l, 48);
  }
  return 0;
}
}

#define CHECK(c, what) do { if (!(c)) { printf("FAILED: %s (line %d; last error: %s)\n", what, __LINE__, sonic::g_msg); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: san_verify <case file>\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
  int64_t hdr[3];
  CHECK(fread(hdr, 8, 3, f) == 3, "header");
  const int64_t n = hdr[0], Q = hdr[1], d = hdr[2];
  auto rd = [&](size_t bytes) { std::vector<uint8_t> v(bytes); if (fread(v.data(), 1, bytes, f) != bytes) v.clear(); return v; };
  std::vector<uint8_t> xb = rd(32), ab = rd(32), wL = rd(32 * Q * n), wR = rd(32 * Q * n), wO = rd(32 * Q * n), cs = rd(32 * Q),
                       proof = rd(sonic_proof_size(Q)), y = rd(32), z = rd(32), yzs = rd(64 * Q), want_pairing = rd(576);
  fclose(f);
  CHECK(!yzs.empty() && !proof.empty() && !want_pairing.empty(), "case file complete");
  sonic_srs srs;
  srs.d = d;
  CHECK(load_fr(xb.data(), srs.x) && load_fr(ab.data(), srs.alpha), "trapdoor canonical");
  int ok = -1;
  CHECK(sonic_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), proof.data(), y.data(), z.data(), yzs.data(), &ok) == 0 && ok == 1, "the golden proof is accepted");
  // one byte changed in one element of every kind: a field element (low byte: stays canonical), a point (leaves the curve)
  const size_t offs[] = {0, 96, 192, 224, 320, 352, 448, 544, 576, 576 + 96, 576 + 128, (size_t)(576 + 224 * Q), (size_t)(576 + 224 * Q + 32),
                         (size_t)(576 + 448 * Q), (size_t)(576 + 448 * Q + 96), proof.size() - 64, proof.size() - 32};
  for (size_t o : offs) {
    std::vector<uint8_t> bad = proof;
    bad[o] ^= 1;
    ok = -1;
    const int rc = sonic_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), bad.data(), y.data(), z.data(), yzs.data(), &ok);
    CHECK((rc == 0 && ok == 0) || rc == SONIC_ERR_BAD_ENCODING, "a tampered proof is rejected");
  }
  {
    std::vector<uint8_t> y2 = y; y2[0] ^= 1;
    CHECK(sonic_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), proof.data(), y2.data(), z.data(), yzs.data(), &ok) == 0 && ok == 0, "other y rejected");
    std::vector<uint8_t> bad = proof;
    memset(&bad[192], 0xff, 32);                                        // prA >= r
    CHECK(sonic_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), bad.data(), y.data(), z.data(), yzs.data(), &ok) == SONIC_ERR_BAD_ENCODING, "non-canonical field element");
    CHECK(sonic_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), nullptr, y.data(), z.data(), yzs.data(), &ok) == SONIC_ERR_INVALID_ARG, "null proof");
  }
  // pcV on its own (CommitmentScheme.hs:51-68): R and its opening at z from the proof: max = n
  CHECK(sonic_pc_v(&srs, n, &proof[0], z.data(), &proof[192], &proof[224], &ok) == 0 && ok == 1, "pcV accepts (R, z, a, W_a)");
  CHECK(sonic_pc_v(&srs, n, &proof[0], y.data(), &proof[192], &proof[224], &ok) == 0 && ok == 0, "pcV rejects another point");
  CHECK(sonic_pc_v(&srs, 3 * d, &proof[0], z.data(), &proof[192], &proof[224], &ok) == SONIC_ERR_SRS_INDEX, "pcV: h^{x^{-d+max}} outside the SRS");
  // hscVerify on the HscProof part
  CHECK(sonic_hsc_verify(&srs, n, Q, wL.data(), wR.data(), wO.data(), Q, yzs.data(), &proof[576], &ok) == 0 && ok == 1, "hscVerify accepts");
  // the Fiat-Shamir verifier on a proof made with drawn challenges: its u, v are not its transcript's -> rejected, no pairing needed
  CHECK(sonic_verify_fs(&srs, n, Q, wL.data(), wR.data(), wO.data(), cs.data(), proof.data(), &ok) == 0 && ok == 0, "verify_fs rejects a proof with foreign challenges");
  // srsPairing (SRS.hs:21,42) against the python oracle's pairing of the same two points
  {
    uint8_t got[576];
    CHECK(sonic_srs_pairing(&srs, got) == 0, "sonic_srs_pairing");
    CHECK(memcmp(got, want_pairing.data(), 576) == 0, "srsPairing equals the oracle's e(g, h^alpha)");
    CHECK(sonic_srs_pairing(nullptr, got) == SONIC_ERR_INVALID_ARG, "null handle");
  }
  printf("san_verify ok\n");
  return 0;
}
