// Opt-in Fiat-Shamir transcript for prove / verify.  The reference's prover DRAWS its challenges (`rnd`, MonadRandom) --
// y after R (src/Sonic/Protocol.hs:66), z after T (:76), y_j / z_j after the openings (:84-85), u after the S_j
// (src/Sonic/Signature.hs:48), v after C and the W'_j / Q_j (:60) -- and hands them to the verifier in RndOracle.  Here each draw
// can instead be a hash of everything that precedes it, in that same order, so that a proof carries its own challenges:
//
//   st_0 = SHA256("sonic-hip/fs/v2" || le64 n || le64 Q || le64 d || circuit digest || srs id)
//   srs id = SHA256("sonic-hip/srs/v1" || le64 d || g^x || g^{alpha x} || g^{1/x} || g^{alpha/x})   (gPositiveX[1], gPositiveAlphaX[0],
//            gNegativeX[0], gNegativeAlphaX[0] of SRS.hs:33-39, 96 canonical bytes each: they determine x and alpha, so a proof is
//            tied to ONE reference string and not to every string of the same d)
//   absorb(label, data):  st <- SHA256(st || label || data)
//   challenge(label, i):  wide = SHA256(st || label || le32 i || 0x00) || SHA256(st || label || le32 i || 0x01)   (64 bytes, little-endian integer)
//                         c = wide mod r, and 1 in place of 0 (evaluation points must be invertible)
//   circuit digest = SHA256("sonic-hip/circuit/v1" || le64 n || le64 Q || wL || wR || wO || cs)    (canonical bytes, row-major)
//
//   absorb("R", R)                                      -> y   = challenge("y", 0)
//   absorb("T", T)                                      -> z   = challenge("z", 0)
//   absorb("open", a || Wa || b || Wb || Wt || s)       -> y_j = challenge("yj", j), z_j = challenge("zj", j)
//   absorb("hscS", [S_j || s_j || W_j]_j)               -> u   = challenge("u", 0)
//   absorb("hscW", C || [s'_j || W'_j || Q_j]_j)        -> v   = challenge("v", 0)
//
// The four blinders c_{n+1..n+4} (Protocol.hs:58) are the prover's secret randomness, not challenges: blinder i =
// wide-reduce(SHA256("sonic-hip/blinder/v2" || seed || circuit digest || srs id || witness digest || le32 i || 0/1)) for a caller-supplied
// 32-byte seed, witness digest = SHA256("sonic-hip/witness/v1" || aL || aR || aO) (canonical bytes).  Statement and witness are mixed in
// the manner of RFC 6979: with the seed alone, two proofs of different assignments under one seed would share their blinders and
// R_1 - R_2 would be an unblinded commitment to the difference of the witnesses.  (v1, round 3, had neither this nor the srs id.)
// The parity tests compare against an independent restatement of this definition with python's hashlib (test infrastructure).
#pragma once
#include <string>
#include "field.hpp"
#include "sha256.hpp"

namespace sonic {

// a - r on the plain 256-bit integers (a >= r)
inline Fr fs_minus_r(const Fr& a) {
  const Fr m = Fr::modulus();
  Fr o;
  uint64_t br = 0;
  for (int i = 0; i < 8; i++) { const uint64_t t = (uint64_t)a.l[i] - m.l[i] - br; o.l[i] = (uint32_t)t; br = (t >> 32) & 1; }
  return o;
}
// 64 little-endian bytes mod r -> canonical (standard-form) Fr
inline Fr fs_wide_reduce(const uint8_t w[64]) {
  Fr lo, hi;
  memcpy(lo.l, w, 32); memcpy(hi.l, w + 32, 32);
  // 2^256 < 3 r: at most two subtractions bring a 256-bit value below r
  for (int k = 0; k < 2; k++) { if (!fp_is_canonical(lo)) lo = fs_minus_r(lo); if (!fp_is_canonical(hi)) hi = fs_minus_r(hi); }
  // hi * 2^256 mod r is exactly the Montgomery encoding of hi
  return fp_add(lo, fp_to_mont(hi));
}

struct FsTranscript {
  uint8_t st[32];
  void init(int64_t n, int64_t Q, int64_t d, const uint8_t digest[32], const uint8_t srs_id[32]) {
    Sha256 h;
    h.update("sonic-hip/fs/v2", 15);
    le64(h, n); le64(h, Q); le64(h, d);
    h.update(digest, 32);
    h.update(srs_id, 32);
    h.finish(st);
  }
  void absorb(const char* label, const uint8_t* data, size_t len) {
    Sha256 h;
    h.update(st, 32); h.update(label, strlen(label)); h.update(data, len);
    h.finish(st);
  }
  void challenge(const char* label, uint32_t i, uint8_t out32[32]) const {
    uint8_t w[64];
    for (uint8_t half = 0; half < 2; half++) {
      Sha256 h;
      h.update(st, 32); h.update(label, strlen(label));
      const uint8_t idx[5] = {(uint8_t)i, (uint8_t)(i >> 8), (uint8_t)(i >> 16), (uint8_t)(i >> 24), half};
      h.update(idx, 5);
      h.finish(w + 32 * half);
    }
    Fr c = fs_wide_reduce(w);
    if (c.is_zero()) c.l[0] = 1;
    memcpy(out32, c.l, 32);
  }
  static void le64(Sha256& h, int64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)((uint64_t)v >> (8 * i)); h.update(b, 8); }
};

// pts: g^x, g^{alpha x}, g^{1/x}, g^{alpha/x} as 4 x 96 canonical bytes
inline void fs_srs_id_of_points(int64_t d, const uint8_t pts[4 * 96], uint8_t out32[32]) {
  Sha256 h;
  h.update("sonic-hip/srs/v1", 16);
  FsTranscript::le64(h, d);
  h.update(pts, 4 * 96);
  h.finish(out32);
}

inline void fs_blinder(const uint8_t seed[32], const uint8_t digest[32], const uint8_t srs_id[32], const uint8_t witness_digest[32], uint32_t i,
                       uint8_t out32[32]) {
  uint8_t w[64];
  for (uint8_t half = 0; half < 2; half++) {
    Sha256 h;
    h.update("sonic-hip/blinder/v2", 20); h.update(seed, 32); h.update(digest, 32); h.update(srs_id, 32); h.update(witness_digest, 32);
    const uint8_t idx[5] = {(uint8_t)i, (uint8_t)(i >> 8), (uint8_t)(i >> 16), (uint8_t)(i >> 24), half};
    h.update(idx, 5);
    h.finish(w + 32 * half);
  }
  Fr c = fs_wide_reduce(w);
  memcpy(out32, c.l, 32);
}

// The challenges a proof determines, in transcript order y, z, y_1..y_Q, z_1..z_Q, u, v (each 32 bytes), from the canonical proof
// bytes (include/sonic_hip.h): what the verifier recomputes.
inline void fs_challenges_of_proof(int64_t n, int64_t Q, int64_t d, const uint8_t digest[32], const uint8_t srs_id[32], const uint8_t* proof,
                                   uint8_t* out) {
  FsTranscript t;
  t.init(n, Q, d, digest, srs_id);
  const uint8_t* R = proof, * T = proof + 96, * open = proof + 192;       // a Wa b Wb Wt s = 32 + 96 + 32 + 96 + 96 + 32 = 384 bytes
  const uint8_t* hscS = proof + 576, * hscW = hscS + Q * 224, * Qv = hscW + Q * 224, * Cc = Qv + 96;
  t.absorb("R", R, 96);
  t.challenge("y", 0, out);
  t.absorb("T", T, 96);
  t.challenge("z", 0, out + 32);
  t.absorb("open", open, 384);
  for (int64_t j = 0; j < Q; j++) { t.challenge("yj", (uint32_t)j, out + 32 * (2 + j)); t.challenge("zj", (uint32_t)j, out + 32 * (2 + Q + j)); }
  t.absorb("hscS", hscS, (size_t)Q * 224);
  t.challenge("u", 0, out + 32 * (2 + 2 * Q));
  std::string w(reinterpret_cast<const char*>(Cc), 96);
  w.append(reinterpret_cast<const char*>(hscW), (size_t)Q * 224);
  t.absorb("hscW", reinterpret_cast<const uint8_t*>(w.data()), w.size());
  t.challenge("v", 0, out + 32 * (3 + 2 * Q));
}

}  // namespace sonic
