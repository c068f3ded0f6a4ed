// AddressSanitizer / UBSan driver for the product's HOST code that needs no GPU (SURVEY section 5 asks for sanitizers on the C++ host
// side; GPU sanitizers are not available on this pool): built by tests/test_sanitizers.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// Covers tests/host/field_host.cpp's entry points (the limb arithmetic and point formulas of field.hpp / g1.hpp as the host compiles
// them), the shared-inversion normalisations of g1_host.hpp, the SHA-256 / Fiat-Shamir transcript code (sha256.hpp, fs.hpp) and the
// SRS file parser (srs_file.hpp) on valid, truncated, padded and hostile files.  Prints "san_host ok".
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "field_host.cpp"
#include "../../sonic_amd/csrc/g1_host.hpp"
#include "../../sonic_amd/csrc/fs.hpp"
#include "../../sonic_amd/csrc/srs_file.hpp"

#define CHECK(c, what) do { if (!(c)) { printf("FAILED: %s (line %d)\n", what, __LINE__); return 1; } } while (0)

static uint64_t st_ = 0x243f6a8885a308d3ull;
static uint64_t rnd64() { st_ ^= st_ << 13; st_ ^= st_ >> 7; st_ ^= st_ << 17; return st_; }

static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v; FILE* f = fopen(p.c_str(), "rb"); if (!f) return v;
  int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v;
}
static void spit(const std::string& p, const std::vector<uint8_t>& v, size_t n) { FILE* f = fopen(p.c_str(), "wb"); if (n) fwrite(v.data(), 1, n, f); fclose(f); }

int main(int argc, char** argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp";
  // ---- field and group entry points of field_host.cpp on random and edge operands ----
  uint8_t a[48], b[48], o[48], g[96], p2[96], p3[96], r1[96], r2[96];
  for (int it = 0; it < 400; it++) {
    for (int i = 0; i < 48; i++) { a[i] = (uint8_t)rnd64(); b[i] = (uint8_t)rnd64(); }
    a[47] &= 0x0f; b[47] &= 0x0f; a[31] &= 0x3f; b[31] &= 0x3f;
    if (it == 0) memset(a, 0, 48);
    for (int op = 0; op < 5; op++) { CHECK(host_fq_op(op, a, b, o) == 0, "host_fq_op"); }
    uint8_t fa[32], fb[32], fo[32];
    memcpy(fa, a, 32); memcpy(fb, b, 32);
    for (int op = 0; op < 5; op++) { CHECK(host_fr_op(op, fa, fb, fo) == 0, "host_fr_op"); }
  }
  CHECK(host_fq_op(9, a, b, o) == -1, "unknown op refused");
  host_gen(g);
  CHECK(host_g1_op(2, g, g, 0, p2) == 0 && host_g1_op(0, p2, g, 0, p3) == 0, "2g, 3g");
  memset(r1, 0, 96);
  for (int op = 0; op < 5; op++) {
    CHECK(host_g1_op(op, p2, p3, 77, r2) == 0, "host_g1_op");
    CHECK(host_g1_op(op, r1, p3, 0, r2) == 0 && host_g1_op(op, p2, r1, 5, r2) == 0 && host_g1_op(op, p3, p3, 3, r2) == 0, "infinity / equal operands");
  }
  CHECK(host_g1_op(4, p2, p3, 0, r2) == 0 && memcmp(r2, p2, 96) == 0, "(p + q) - q == p");
  {
    // shared-inversion normalisations, with points at infinity and the empty batch
    std::vector<G1XYZZ> pts;
    G1Affine gen = load_pt(g);
    G1XYZZ acc = g1_dbl_affine(gen);
    for (int i = 0; i < 33; i++) { acc = g1_add_mixed(g1_dbl(acc), gen); pts.push_back(i % 5 == 2 ? G1XYZZ::inf() : acc); }
    std::vector<uint8_t> one(96 * pts.size()), all(96 * pts.size());
    for (size_t i = 0; i < pts.size(); i++) g1_canonical_bytes_host(pts[i], &one[96 * i]);
    g1_canonical_bytes_host_batch(pts.data(), (int)pts.size(), all.data());
    CHECK(one == all, "batch canonical bytes");
    std::vector<G1Affine> aff(pts.size());
    g1_batch_affine_host(pts.data(), (long)pts.size(), aff.data());
    g1_canonical_bytes_host_batch(pts.data(), 0, all.data());
    g1_batch_affine_host(pts.data(), 0, aff.data());
  }
  // ---- SHA-256 over every length around the block boundaries, split updates; the transcript ----
  {
    std::vector<uint8_t> msg(300);
    for (auto& c : msg) c = (uint8_t)rnd64();
    for (size_t n = 0; n <= msg.size(); n++) {
      uint8_t d1[32], d2[32];
      Sha256 h1; h1.update(msg.data(), n); h1.finish(d1);
      Sha256 h2; const size_t cut = n / 3; h2.update(msg.data(), cut); h2.update(msg.data() + cut, 0); h2.update(msg.data() + cut, n - cut); h2.finish(d2);
      CHECK(memcmp(d1, d2, 32) == 0, "sha256: split update == one update");
    }
    uint8_t abc[32];
    Sha256 h; h.update("abc", 3); h.finish(abc);
    static const uint8_t want[32] = {0xba, 0x78, 0x16, 0xbf, 0x8f, 0x01, 0xcf, 0xea, 0x41, 0x41, 0x40, 0xde, 0x5d, 0xae, 0x22, 0x23,
                                     0xb0, 0x03, 0x61, 0xa3, 0x96, 0x17, 0x7a, 0x9c, 0xb4, 0x10, 0xff, 0x61, 0xf2, 0x00, 0x15, 0xad};
    CHECK(memcmp(abc, want, 32) == 0, "sha256(\"abc\") (FIPS 180-4)");
    uint8_t w[64];
    memset(w, 0xff, 64);
    Fr top = fs_wide_reduce(w);                                   // 2^512 - 1 mod r: both halves need two subtractions
    CHECK(fp_is_canonical(top), "wide reduction of 2^512 - 1 is canonical");
    memset(w, 0, 64);
    CHECK(fs_wide_reduce(w).is_zero(), "wide reduction of 0");
    const int64_t Q = 3;
    std::vector<uint8_t> proof((7 + 4 * Q) * 96 + (5 + 2 * Q) * 32), ch(32 * (4 + 2 * Q));
    for (auto& c : proof) c = (uint8_t)rnd64();
    uint8_t dg[32] = {1, 2, 3};
    uint8_t sid[32], pts[4 * 96];
    for (auto& c : pts) c = (uint8_t)rnd64();
    fs_srs_id_of_points(61, pts, sid);
    fs_challenges_of_proof(8, Q, 61, dg, sid, proof.data(), ch.data());
    for (int i = 0; i < 4 + 2 * Q; i++) { Fr c; memcpy(c.l, &ch[32 * i], 32); CHECK(fp_is_canonical(c) && !c.is_zero(), "challenge canonical and non-zero"); }
    uint8_t bl[32];
    for (uint32_t i = 0; i < 4; i++) { fs_blinder(dg, dg, sid, sid, i, bl); Fr c; memcpy(c.l, bl, 32); CHECK(fp_is_canonical(c), "blinder canonical"); }
  }
  // ---- SRS file parser: a valid file (with and without the G2 half), every truncation class, trailing bytes, hostile headers ----
  {
    const int64_t d = 3, n = 2 * d + 1;
    for (int with_g2 = 0; with_g2 < 2; with_g2++) {
      const std::string path = tmp + "/san_srs.bin";
      FILE* f = fopen(path.c_str(), "wb");
      CHECK(f && srs_file_write_header(f, d, with_g2 != 0), "write header");
      std::vector<uint8_t> body((size_t)(2 * n * 96 + (with_g2 ? 2 * n * 192 : 0)));
      for (auto& c : body) c = (uint8_t)rnd64();
      fwrite(body.data(), 1, body.size(), f); fclose(f);
      SrsFile s; std::string err;
      CHECK(srs_file_read(path.c_str(), s, err) == 0 && s.d == d && s.has_g2() == (with_g2 != 0) && s.g0.size() == (size_t)n * 96 &&
            s.h1.size() == (with_g2 ? (size_t)n * 192 : 0) && memcmp(s.g1.data(), body.data() + n * 96, (size_t)n * 96) == 0, "valid file parses");
      const std::vector<uint8_t> whole = slurp(path);
      const std::string bad = tmp + "/san_srs_bad.bin";
      for (size_t cut : {(size_t)0, (size_t)7, (size_t)8, (size_t)23, (size_t)24, (size_t)25, whole.size() / 2, whole.size() - 1}) {
        spit(bad, whole, cut);
        SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) != 0 && t.g0.empty(), "truncated file refused");
      }
      std::vector<uint8_t> more = whole; more.push_back(0);
      spit(bad, more, more.size());
      { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 3, "trailing byte refused"); }
      auto with = [&](size_t off, std::initializer_list<uint8_t> bytes) { std::vector<uint8_t> v = whole; size_t k = off; for (uint8_t c : bytes) v[k++] = c; spit(bad, v, v.size()); };
      with(0, {'X'});                                                    { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "bad magic"); }
      with(8, {3, 0, 0, 0});                                             { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "unknown version"); }
      with(12, {2, 0, 0, 0});                                            { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "unknown flag bits"); }
      with(16, {0, 0, 0, 0, 0, 0, 0, 0});                                { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "d = 0"); }
      with(16, {0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff});        { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "d = -1"); }
      // d just under the cap: 2^40 points promised by a 1-KB file -- refused by the size check, nothing of that size is allocated
      with(16, {0xff, 0xff, 0xff, 0xff, 0x7f, 0, 0, 0});                 { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 3 && t.g0.empty(), "huge d against a small file"); }
      with(16, {4, 0, 0, 0, 0, 0, 0, 0});                                { SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 3, "d that does not match the size"); }
      if (with_g2) { with(8, {1, 0, 0, 0}); SrsFile t; CHECK(srs_file_read(bad.c_str(), t, err) == 2, "version 1 cannot carry the G2 flag"); }
    }
    SrsFile t; std::string err;
    CHECK(srs_file_read((tmp + "/does_not_exist.bin").c_str(), t, err) == 1, "missing file");
  }
  printf("san_host ok\n");
  return 0;
}
