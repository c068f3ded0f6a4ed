// Host-side normalisation of G1 results (no HIP in here: tests/pairing_selftest.cpp compiles it with g++): canonical bytes of
// one XYZZ point, and of n points with one shared inversion (Montgomery's trick).
#pragma once
#include <string.h>
#include <vector>
#include "g1.hpp"

namespace sonic {

inline void g1_canonical_bytes_host(const G1XYZZ& p, uint8_t* out) {
  G1Affine a = g1_to_affine(p);
  uint32_t w[24];
  if (a.is_inf()) { for (int i = 0; i < 24; i++) w[i] = 0; }
  else {
    Fq x = fp_from_mont(a.x), y = fp_from_mont(a.y);
    for (int i = 0; i < 12; i++) { w[i] = x.l[i]; w[12 + i] = y.l[i]; }
  }
  memcpy(out, w, 96);
}

// Montgomery-form affine points from XYZZ ones with one shared inversion (host; tables built at prepare time)
inline void g1_batch_affine_host(const G1XYZZ* p, long n, G1Affine* out) {
  std::vector<Fq> den((size_t)n), pre((size_t)n);
  Fq acc = Fq::one();
  for (long i = 0; i < n; i++) {
    den[i] = p[i].is_inf() ? Fq::one() : fp_mul(p[i].zz, p[i].zzz);
    pre[i] = acc;
    acc = fp_mul(acc, den[i]);
  }
  Fq inv = fp_inv(acc);
  for (long i = n - 1; i >= 0; i--) {
    const Fq di = fp_mul(inv, pre[i]);
    inv = fp_mul(inv, den[i]);
    if (p[i].is_inf()) { out[i] = G1Affine::inf(); continue; }
    out[i].x = fp_mul(p[i].x, fp_mul(di, p[i].zzz));
    out[i].y = fp_mul(p[i].y, fp_mul(di, p[i].zz));
  }
}

// n points at once: one inversion for all of them (Montgomery's trick), ~10 products per point -- a proof's 7 + 4Q results
// normalise in ~40 us on the calling thread instead of one inversion each on threads of their own
inline void g1_canonical_bytes_host_batch(const G1XYZZ* p, int n, uint8_t* out) {
  std::vector<Fq> den((size_t)n), pre((size_t)n);
  Fq acc = Fq::one();
  for (int i = 0; i < n; i++) {
    den[i] = p[i].is_inf() ? Fq::one() : fp_mul(p[i].zz, p[i].zzz);
    pre[i] = acc;
    acc = fp_mul(acc, den[i]);
  }
  Fq inv = fp_inv(acc);
  for (int i = n - 1; i >= 0; i--) {
    const Fq di = fp_mul(inv, pre[i]);                 // 1 / (zz zzz) of point i
    inv = fp_mul(inv, den[i]);
    uint32_t w[24];
    if (p[i].is_inf()) { for (int k = 0; k < 24; k++) w[k] = 0; }
    else {
      const Fq x = fp_from_mont(fp_mul(p[i].x, fp_mul(di, p[i].zzz))), y = fp_from_mont(fp_mul(p[i].y, fp_mul(di, p[i].zz)));
      for (int k = 0; k < 12; k++) { w[k] = x.l[k]; w[12 + k] = y.l[k]; }
    }
    memcpy(out + 96 * (size_t)i, w, 96);
  }
}

}  // namespace sonic
