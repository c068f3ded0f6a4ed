// Host (g++) build of the product's limb arithmetic and point formulas, for CPU unit tests.
// The same headers are compiled by hipcc for gfx950; this only exercises their logic on the host.
#include <string.h>
#include "../../sonic_amd/csrc/g1.hpp"
#include "../../sonic_amd/csrc/endo.hpp"
using namespace sonic;

template <class F> static F load(const uint8_t* b) { F a; memcpy(a.l, b, sizeof a.l); return a; }
template <class F> static void store(uint8_t* b, const F& a) { memcpy(b, a.l, sizeof a.l); }

extern "C" {
// op: 0 mul, 1 add, 2 sub, 3 neg(a), 4 inv(a); canonical standard-form little-endian bytes in/out
int host_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fq x = fp_to_mont(load<Fq>(a)), y = fp_to_mont(load<Fq>(b)), r;
  switch (op) { case 0: r = fp_mul(x, y); break; case 1: r = fp_add(x, y); break; case 2: r = fp_sub(x, y); break;
    case 3: r = fp_neg(x); break; case 4: r = fp_inv(x); break; default: return -1; }
  store(out, fp_from_mont(r)); return 0;
}
int host_fr_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fr x = fp_to_mont(load<Fr>(a)), y = fp_to_mont(load<Fr>(b)), r;
  switch (op) { case 0: r = fp_mul(x, y); break; case 1: r = fp_add(x, y); break; case 2: r = fp_sub(x, y); break;
    case 3: r = fp_neg(x); break; case 4: r = fp_inv(x); break; default: return -1; }
  store(out, fp_from_mont(r)); return 0;
}
static G1Affine load_pt(const uint8_t* b) { G1Affine p; p.x = load<Fq>(b); p.y = load<Fq>(b + 48);
  if (p.is_inf()) return p; p.x = fp_to_mont(p.x); p.y = fp_to_mont(p.y); return p; }
static void store_pt(uint8_t* b, const G1Affine& p) { if (p.is_inf()) { memset(b, 0, 96); return; }
  store(b, fp_from_mont(p.x)); store(b + 48, fp_from_mont(p.y)); }
// op: 0 mixed add (xyzz(a) + affine b), 1 full add, 2 dbl(a), 3 a*k (k small), 4 ((a+b)+b)-style chain exercising xyzz state
int host_g1_op(int op, const uint8_t* a, const uint8_t* b, uint32_t k, uint8_t* out) {
  G1Affine p = load_pt(a), q = load_pt(b);
  G1XYZZ r;
  switch (op) {
    case 0: r = g1_add_mixed(G1XYZZ::from_affine(p), q); break;
    case 1: { G1XYZZ pp = g1_dbl(g1_dbl_affine(p)); G1XYZZ qq = g1_add_mixed(g1_dbl_affine(q), q);  // 4p + 3q with non-trivial zz
              r = g1_add(pp, qq); break; }
    case 2: r = g1_dbl(G1XYZZ::from_affine(p)); break;
    case 3: r = g1_mul_small(g1_add_mixed(g1_dbl_affine(p), p), k); break;  // k * 3p
    case 4: r = g1_add(g1_add_mixed(G1XYZZ::from_affine(p), q), g1_neg(G1XYZZ::from_affine(q))); break;  // (p+q)-q
    default: return -1;
  }
  store_pt(out, g1_to_affine(r)); return 0;
}
// the endomorphism split s = s1 + lambda s2 (endo.hpp): standard-form bytes in and out
int host_endo_split(const uint8_t* s, uint8_t* s1, uint8_t* s2) {
  Fr a, b;
  endo_split(load<Fr>(s), a, b);
  store(s1, a); store(s2, b); return 0;
}
// phi(P) = (beta x, y), through the XYZZ form with a non-trivial ZZ (2P)
int host_endo_phi(const uint8_t* a, uint8_t* out) {
  store_pt(out, g1_to_affine(g1_endo(g1_dbl_affine(load_pt(a))))); return 0;
}
int host_gen(uint8_t* out) { G1Affine g; uint32_t gx[12] = G1_GEN_X_MONT, gy[12] = G1_GEN_Y_MONT;
  memcpy(g.x.l, gx, 48); memcpy(g.y.l, gy, 48); store_pt(out, g); return 0; }
}
