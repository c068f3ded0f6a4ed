// ONE proof over `world` GPUs: which part of which MSM a rank runs (host only, no HIP; also compiled by tests/host).
//
// The 7 + 4Q commitments and openings of prove + hscProve (src/Sonic/Protocol.hs:63,73,79-81; src/Sonic/Signature.hs:40-45,
// 51-57,63) are independent sums once the transcript is known.  Every rank holds circuit, assignment, transcript and SRS
// (replicated), builds only the polynomials its MSMs read, and runs a contiguous piece of the "work line": the MSMs laid end to
// end, group by group (the MSMs of a group read the same polynomial), cut so that the slowest rank is as fast as possible.  A cut
// may fall inside an MSM: its term range is then split and both ranks contribute a partial sum.  Cost model of a rank, in units of
// one MSM term: the terms it runs + a fixed part per MSM piece (the bucket reduction does not shrink with the term range) + what
// the polynomials it has to build cost (above all the t(X,y) product, which every rank that touches T or W_t repeats).
//
// The plan is a pure function of (n, Q, prepared, world, cost constants): every rank computes all of it and reads its own row.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace sonic {

constexpr uint32_t SHARE_ONE = 1u << 20;      // a slot's term range in units of 1 / 2^20: [lo, hi) of [0, SHARE_ONE]
struct SlotShare { uint32_t lo = 0, hi = 0; };

// polynomials a group of MSMs reads (bits)
enum { POLY_R1 = 1, POLY_SY0 = 2, POLY_T = 4, POLY_SU = 8, POLY_SYJ0 = 16 /* << j, j < 27; beyond that all j share one bit */ };

struct ShareCosts {
  // term-equivalents; defaults from the least-squares fit of tools/prove_strong.py --fit on one MI355X at n = 2^20, Q = 2
  // (profiles/r04_prove_strong_emulated.txt: 2.7 ms per n terms, ~0.2 ms per MSM piece whatever its size -- sort launches and the
  // reduction tree of 2^19 buckets: 2 x NB / W terms -- the t product 4.2-5.0 ms, the other polynomials inside the noise)
  double per_job_buckets = 2.0;     // x NB / W terms: the fixed part of one MSM piece
  double r1 = 0.02, sy = 0.05, su = 0.05, tprod = 1.8;    // x n terms: build_r1, s(X,y), s(u,Y), the t(X,y) product incl. its operands
  static double env(const char* name, double dflt) { const char* s = getenv(name); return s && *s ? atof(s) : dflt; }
  static ShareCosts from_env() {
    ShareCosts c;
    c.per_job_buckets = env("SONIC_SHARE_COST_JOB", c.per_job_buckets);
    c.r1 = env("SONIC_SHARE_COST_R1", c.r1); c.sy = env("SONIC_SHARE_COST_SY", c.sy);
    c.su = env("SONIC_SHARE_COST_SU", c.su); c.tprod = env("SONIC_SHARE_COST_T", c.tprod);
    return c;
  }
};

struct SharePlan {
  long n = 0, Q = 0;
  int world = 1;
  int K = 0;                                   // 7 + 4Q slots (proof order of prove.hip: R, T, Wa, Wb, Wt, [S_j, W_j], [W'_j, Q_j], Qv, C)
  std::vector<SlotShare> rows;                 // world x K
  std::vector<uint32_t> polys;                 // per rank: POLY_* it has to build
  std::vector<double> cost;                    // per rank, modelled (terms)
  const SlotShare* row(int rank) const { return &rows[(size_t)rank * K]; }
  bool owns(int rank, int slot) const { const SlotShare& s = rows[(size_t)rank * K + slot]; return s.hi > s.lo; }
  // the rank that reports an opening's evaluation / runs a slot's side jobs: the one whose piece starts at term 0
  bool first_piece(int rank, int slot) const { const SlotShare& s = rows[(size_t)rank * K + slot]; return s.hi > s.lo && s.lo == 0; }
};

struct ShareItem { int slot; long terms; uint32_t polys; };

// the work line: groups in an order that keeps shared polynomials on neighbouring ranks (T and W_t need r(X,1) like R, W_a, W_b)
inline std::vector<ShareItem> share_line(long n, long Q, bool prepared) {
  std::vector<ShareItem> it;
  auto pj = [](long j) { return (uint32_t)POLY_SYJ0 << (j < 27 ? j : 27); };
  it.push_back({1, 7 * n + 9, POLY_R1 | POLY_SY0 | POLY_T});                       // T
  it.push_back({4, 7 * n + 8, POLY_R1 | POLY_SY0 | POLY_T});                       // W_t
  it.push_back({0, 3 * n + 4, POLY_R1});                                           // R (the X^0 coefficient of r is zero)
  it.push_back({2, 3 * n + 4, POLY_R1});                                           // W_a
  it.push_back({3, 3 * n + 4, POLY_R1});                                           // W_b
  for (long j = 0; j < Q; j++) {
    it.push_back({(int)(5 + 2 * j), prepared ? n : 3 * n + 1, pj(j)});             // S_j
    it.push_back({(int)(6 + 2 * j), 3 * n, pj(j)});                                // W_j
    it.push_back({(int)(5 + 2 * Q + 2 * j), 3 * n, pj(j)});                        // W'_j
  }
  it.push_back({(int)(6 + 4 * Q), 2 * n + Q + 1, POLY_SU});                        // C
  for (long j = 0; j < Q; j++) it.push_back({(int)(6 + 2 * Q + 2 * j), 2 * n + Q, POLY_SU});   // Q_j
  it.push_back({(int)(5 + 4 * Q), 2 * n + Q, POLY_SU});                            // Q_v
  return it;
}

inline double share_poly_cost(uint32_t polys, long n, const ShareCosts& c) {
  double f = 0;
  if (polys & POLY_R1) f += c.r1 * n;
  if (polys & POLY_SY0) f += c.sy * n;
  if (polys & POLY_T) f += c.tprod * n;
  if (polys & POLY_SU) f += c.su * n;
  for (uint32_t b = POLY_SYJ0; b; b <<= 1) if (polys & b) f += c.sy * n;
  return f;
}

// Greedy fill for a makespan L: ranks take the line in order; returns the number of ranks used (world + 1: does not fit) and,
// if `out` is given, writes the cut positions (item index, terms taken of it) per rank.
struct ShareCut { int item; long from, to; };
inline int share_fill(const std::vector<ShareItem>& line, long n, double job_fixed, const ShareCosts& c, double L, int world,
                      std::vector<std::vector<ShareCut>>* out) {
  int rank = 0;
  double cost = 0;
  uint32_t polys = 0;
  if (out) { out->assign((size_t)world, {}); }
  for (size_t i = 0; i < line.size(); i++) {
    long done = 0;
    while (done < line[i].terms) {
      if (rank >= world) return world + 1;
      const double enter = share_poly_cost(polys | line[i].polys, n, c) - share_poly_cost(polys, n, c) + job_fixed;
      const double room = L - cost - enter;
      // a piece must be worth its fixed part: at least as many terms as the fixed part costs, or the rest of the item
      const long rest = line[i].terms - done;
      const long min_piece = std::min<long>(rest, (long)(enter) + 1);
      if (room < (double)min_piece) {
        if (cost == 0) return world + 1;      // an empty rank cannot take even the smallest piece: L is too small
        rank++; cost = 0; polys = 0;
        continue;
      }
      const long take = std::min<long>(rest, (long)room);
      if (out) (*out)[(size_t)rank].push_back({(int)i, done, done + take});
      cost += enter + (double)take;
      polys |= line[i].polys;
      done += take;
    }
  }
  return rank + 1;
}

// NB, W: buckets per set and windows of the MSM plan the proof's MSMs run with (the reduction of a piece costs ~ per_job_buckets x NB
// full additions against W additions per term)
inline SharePlan share_plan(long n, long Q, bool prepared, int world, long NB, int W, const ShareCosts& c = ShareCosts()) {
  SharePlan pl;
  pl.n = n; pl.Q = Q; pl.world = world; pl.K = (int)(7 + 4 * Q);
  pl.rows.assign((size_t)world * pl.K, SlotShare());
  pl.polys.assign((size_t)world, 0);
  pl.cost.assign((size_t)world, 0.0);
  const std::vector<ShareItem> line = share_line(n, Q, prepared);
  const double job_fixed = c.per_job_buckets * (double)NB / (double)(W > 0 ? W : 1);
  double total = 0;
  for (auto& it : line) total += (double)it.terms + job_fixed;
  double lo = total / world, hi = total + share_poly_cost(~0u, n, c) * 2 + job_fixed * 4 + 16;     // hi always fits on one rank
  for (int iter = 0; iter < 60 && hi - lo > 1.0; iter++) {
    const double mid = 0.5 * (lo + hi);
    if (share_fill(line, n, job_fixed, c, mid, world, nullptr) <= world) hi = mid; else lo = mid;
  }
  std::vector<std::vector<ShareCut>> cuts;
  share_fill(line, n, job_fixed, c, hi, world, &cuts);
  // term cuts -> fractions of the slot in units of 1 / SHARE_ONE.  Neighbouring pieces share the boundary value, so the pieces of
  // a slot partition [0, SHARE_ONE] whatever the rounding.
  auto frac = [](long pos, long terms) { return (uint32_t)(((unsigned __int128)pos * SHARE_ONE) / (unsigned long)terms); };
  for (int r = 0; r < world; r++) {
    for (const ShareCut& ct : cuts[(size_t)r]) {
      const ShareItem& it = line[(size_t)ct.item];
      SlotShare& s = pl.rows[(size_t)r * pl.K + it.slot];
      s.lo = ct.from == 0 ? 0 : frac(ct.from, it.terms);
      s.hi = ct.to == it.terms ? SHARE_ONE : frac(ct.to, it.terms);
      if (s.hi > s.lo) {
        pl.polys[(size_t)r] |= it.polys;
        pl.cost[(size_t)r] += job_fixed + (double)(ct.to - ct.from);
      }
    }
    pl.cost[(size_t)r] += share_poly_cost(pl.polys[(size_t)r], n, c);
  }
  return pl;
}

// piece [lo, hi) of SHARE_ONE -> term range of a job with `terms` terms; consistent on every rank (same integer arithmetic)
inline void share_term_range(const SlotShare& s, long terms, long* t0, long* t1) {
  auto at = [&](uint32_t f) { return f >= SHARE_ONE ? terms : (long)(((unsigned __int128)terms * f) >> 20); };
  *t0 = at(s.lo); *t1 = at(s.hi);
}

}  // namespace sonic
