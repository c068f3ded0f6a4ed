// On-disk SRS container (host only; no HIP): "SONICSRS" | u32 version (1 | 2) | u32 flags (bit 0: G2 half follows; version 2 only) |
// i64 d | basis 0, basis 1 as (2d+1) x 96 canonical bytes | [G2 basis 0, basis 1 as (2d+1) x 192 bytes].  The reference has no
// persistence (src/Sonic/SRS.hs builds the SRS in memory); this is the container sonic_srs_save / sonic_srs_load speak.  Kept
// apart from the device code so that the parsing of an untrusted file is unit-tested under AddressSanitizer / UBSan on the host
// (tests/host/san_host.cpp): the header is checked against the file's real size BEFORE anything is allocated from it.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace sonic {

constexpr char SRS_FILE_MAGIC[8] = {'S', 'O', 'N', 'I', 'C', 'S', 'R', 'S'};
constexpr int64_t SRS_FILE_MAX_D = 1LL << 40;

struct SrsFile {
  uint32_t version = 0, flags = 0;
  int64_t d = 0;
  std::vector<uint8_t> g0, g1, h0, h1;      // canonical bytes; h0 / h1 empty without the G2 half
  bool has_g2() const { return (flags & 1u) != 0; }
};

inline bool srs_file_write_header(FILE* f, int64_t d, bool with_g2) {
  const uint32_t ver = 2, flags = with_g2 ? 1u : 0u;
  return fwrite(SRS_FILE_MAGIC, 1, 8, f) == 8 && fwrite(&ver, 4, 1, f) == 1 && fwrite(&flags, 4, 1, f) == 1 && fwrite(&d, 8, 1, f) == 1;
}

// 0 on success; otherwise a message in `err` (nothing is kept of a file that fails)
inline int srs_file_read(const char* path, SrsFile& out, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("cannot open ") + path; return 1; }
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  char magic[8];
  uint32_t ver = 0, flags = 0;
  int64_t d = 0;
  const bool head = fread(magic, 1, 8, f) == 8 && memcmp(magic, SRS_FILE_MAGIC, 8) == 0 && fread(&ver, 4, 1, f) == 1 && fread(&flags, 4, 1, f) == 1 &&
                    fread(&d, 8, 1, f) == 1 && (ver == 1 || ver == 2) && d >= 1 && d < SRS_FILE_MAX_D && (flags & ~1u) == 0 && !(ver == 1 && flags);
  if (!head) { err = std::string(path) + " is not a version-1/2 SRS file"; return 2; }
  // the sizes the header promises against the size the file has, before any allocation
  if (fseek(f, 0, SEEK_END) != 0) { err = "cannot seek"; return 3; }
  const long long actual = ftell(f);
  const unsigned long long n = 2ull * (unsigned long long)d + 1ull;
  const unsigned long long expect = 24ull + 2ull * n * 96ull + ((flags & 1u) ? 2ull * n * 192ull : 0ull);
  if (actual < 0 || (unsigned long long)actual != expect) { err = std::string(path) + " is truncated or has trailing bytes"; return 3; }
  if (fseek(f, 24, SEEK_SET) != 0) { err = "cannot seek"; return 3; }
  SrsFile s;
  s.version = ver; s.flags = flags; s.d = d;
  s.g0.resize(96 * (size_t)n); s.g1.resize(96 * (size_t)n);
  bool ok = fread(s.g0.data(), 96, (size_t)n, f) == (size_t)n && fread(s.g1.data(), 96, (size_t)n, f) == (size_t)n;
  if (ok && (flags & 1u)) {
    s.h0.resize(192 * (size_t)n); s.h1.resize(192 * (size_t)n);
    ok = fread(s.h0.data(), 192, (size_t)n, f) == (size_t)n && fread(s.h1.data(), 192, (size_t)n, f) == (size_t)n;
  }
  if (ok) ok = fgetc(f) == EOF;
  if (!ok) { err = std::string(path) + " is truncated or has trailing bytes"; return 3; }
  out = std::move(s);
  return 0;
}

}  // namespace sonic
