// Path selection for the path index: SeedFinder::pick_paths (reference
// include/psi/seed_finder.hpp:1138-1167) -- `n` haplotype-like walks per embedded path, drawn by the
// Haplotyper graph iterator, as full paths or cut into patches.
//
//   Haplotyper (include/psi/graph_iter.hpp:537-731)        ->  class Haplotyper below: the same
//       choice of the next node, rule for rule (first out-edge at level 0; at level s > 0 the
//       first out-edge whose "setback" path -- the last nodes of the walk, kept short by the
//       entropy rule -- no earlier walk contains; else the out-edge whose setback path the fewest
//       earlier walks contain; else a uniformly random out-edge).  The reference draws that last
//       case from a randomly seeded generator; here the generator is seeded by the caller.
//   get_uniq_full_haplotype (pathindex.hpp:455-496)        ->  full_walk()
//   get_uniq_patches (pathindex.hpp:498-560)               ->  cut_patches(): own formulation of the
//       same idea.  A walk drawn after the first one is mostly a repetition of earlier walks; only
//       its windows of `context` bases whose NODE sequence no earlier walk contains are new.  The
//       patches are the maximal runs of such windows, trimmed to the base: a patch starts at the
//       first base of its first uncovered window (head offset in its first node, Path::left) and ends
//       with the last base of its last one (tail length in its last node, Path::right).  The
//       reference slides a frontier path along the walk and trims with ltrim/rtrim_front_by_len;
//       its patches can differ from these by the context carried at their ends.  Which bases the
//       index holds does not change the hit set: every k-walk the indexed text does not spell
//       makes its first base a starting locus (find_starting_loci, index.cpp).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <random>
#include <unordered_map>

#include "host.hpp"

namespace psigpu {
namespace {

constexpr uint32_t END = 0xFFFFFFFFu;

// An earlier walk, kept the way Path<Graph, Haplotype> keeps it: the set of its node ranks.
// contains(q) (path_base.hpp:1256-1280): q's ranks strictly increase, all lie on the walk, and the walk
// has no other node between q's first and last rank -- q is a contiguous run of the walk.
struct Walk {
  std::vector<uint32_t> sorted;       // node ranks, increasing
  std::vector<uint64_t> bits;         // the same set as a bit per node (the reference's Haplotype path is a bit vector too)
  bool bit(uint32_t v) const { return (bits[v >> 6] >> (v & 63)) & 1ull; }
  bool contains(const uint32_t* q, size_t n) const
  {
    if (n == 0) return false;
    const uint32_t b = q[0], e = q[n - 1];
    if (e < b) return false;
    if (!bits.empty() && e - b < 4096) {
      // the usual case, ranks close together: q is the walk's node set in [b, e] iff its ranks increase, all
      // lie on the walk and the walk has exactly n nodes there -- no binary search over a 300 M-node walk
      if (!bit(b) || !bit(e)) return false;
      for (size_t i = 1; i < n; ++i)
        if (q[i] <= q[i - 1] || !bit(q[i])) return false;
      uint64_t cnt = 0;
      const uint32_t wb = b >> 6, we = e >> 6;
      for (uint32_t w = wb; w <= we; ++w) {
        uint64_t x = bits[w];
        if (w == wb) x &= ~0ull << (b & 63);
        if (w == we) x &= ~0ull >> (63 - (e & 63));
        cnt += (uint64_t)__builtin_popcountll(x);
      }
      return cnt == n;
    }
    auto lo = std::lower_bound(sorted.begin(), sorted.end(), b);
    auto hi = std::upper_bound(sorted.begin(), sorted.end(), e);
    if (lo == sorted.end() || *lo != b || (size_t)(hi - lo) != n) return false;
    for (size_t i = 0; i < n; ++i)
      if (lo[i] != q[i]) return false;      // (strictly increasing and on the walk, both at once)
    return true;
  }
};

class Haplotyper {
public:
  Haplotyper(const Graph& g, uint64_t seed) : g_(g), rng_((uint32_t)(seed ^ (seed >> 32)) + 5489u) {}

  void reset(uint32_t start)                       // GraphIter::reset (:706-718)
  {
    start_ = value_ = start;
    visiting_.clear(); visited_.clear();
    current_.assign(1, start);
    setback_ = 0; entropy_ = 1;
  }
  uint32_t value() const { return value_; }
  size_t level() const { return visited_.size(); } // :720-724
  const std::vector<uint32_t>& current() const { return current_; }

  bool covered(const uint32_t* q, size_t n) const  // operator[] (:693-700): covered_by( path, visited )
  {
    for (const Walk& w : visited_) if (w.contains(q, n)) return true;
    return false;
  }

  void next()                                      // operator++ (:590-657)
  {
    const uint64_t e0 = g_.edge_off[value_], e1 = g_.edge_off[value_ + 1];
    if (e0 == e1) { value_ = END; return; }
    if (setback_ > 1) {
      while (!visiting_.empty() && entropy_ > setback_) {
        entropy_ /= outdeg(visiting_.front());
        visiting_.pop_front();
      }
    }
    uint32_t candidate = END;
    if (setback_ == 0 || e1 - e0 == 1) candidate = g_.edge_to[e0];
    else {
      // a forward node such that the setback path is in none of the earlier walks
      bool again;
      do {
        again = false;
        for (uint64_t e = e0; e < e1; ++e) {
          visiting_.push_back(g_.edge_to[e]);
          const bool seen = covered_visiting();
          visiting_.pop_back();
          if (!seen) { candidate = g_.edge_to[e]; break; }
        }
        if (setback_ == 1 && candidate == END && visiting_.empty()) { visiting_.push_back(value_); again = true; }
      } while (again);
      if (setback_ == 1 && !visiting_.empty()) visiting_.pop_back();
    }
    if (candidate == END) candidate = least_covered_adjacent();       // graph.hpp:250-284
    if (candidate == END) {                                           // graph.hpp:162-203
      std::uniform_int_distribution<uint64_t> dis(0, e1 - e0 - 1);
      candidate = g_.edge_to[e0 + dis(rng_)];
    }
    value_ = candidate;
    if (setback_ > 1) { visiting_.push_back(value_); entropy_ *= std::max<uint64_t>(1, outdeg(value_)); }
    current_.push_back(value_);
  }

  void discard() { rewind(); }                     // operator--(int) (:659-671)
  void save()                                      // operator--() (:673-680)
  {
    Walk w;
    w.sorted = current_;
    // (a walk through a graph whose ranks follow the topology -- vg's do -- is increasing already: the sort of a
    // 300 M-node walk was a fifth of a whole-genome index build)
    bool increasing = true;
    for (size_t i = 1; i < w.sorted.size() && increasing; ++i) increasing = w.sorted[i - 1] < w.sorted[i];
    if (!increasing) {
      std::sort(w.sorted.begin(), w.sorted.end());
      w.sorted.erase(std::unique(w.sorted.begin(), w.sorted.end()), w.sorted.end());
    }
    w.bits.assign((g_.n_nodes() >> 6) + 1, 0);
    for (uint32_t v : w.sorted) w.bits[v >> 6] |= 1ull << (v & 63);
    visited_.push_back(std::move(w));
    setback_ = (unsigned)visited_.size();
    rewind();
  }

private:
  uint64_t outdeg(uint32_t v) const { return g_.edge_off[v + 1] - g_.edge_off[v]; }
  bool covered_visiting() const
  {
    tmp_.assign(visiting_.begin(), visiting_.end());
    return covered(tmp_.data(), tmp_.size());
  }
  void rewind()
  {
    value_ = start_;
    visiting_.clear();
    entropy_ = 1;
    if (setback_ > 1) { visiting_.push_back(value_); entropy_ *= std::max<uint64_t>(1, outdeg(value_)); }
    current_.assign(1, value_);
  }
  uint32_t least_covered_adjacent()
  {
    if (visiting_.empty()) return END;
    const uint32_t back = visiting_.back();
    uint32_t lc = END;
    uint64_t lc_value = ~0ull;
    bool equal = true;
    for (uint64_t e = g_.edge_off[back]; e < g_.edge_off[back + 1]; ++e) {
      visiting_.push_back(g_.edge_to[e]);
      tmp_.assign(visiting_.begin(), visiting_.end());
      visiting_.pop_back();
      uint64_t cov = 0;
      for (const Walk& w : visited_) cov += w.contains(tmp_.data(), tmp_.size());
      if (equal && lc_value != ~0ull && lc_value != cov) equal = false;
      if (cov < lc_value) { lc = g_.edge_to[e]; lc_value = cov; }
    }
    return equal ? END : lc;
  }

  const Graph& g_;
  std::mt19937 rng_;
  uint32_t start_ = 0, value_ = 0;
  std::deque<uint32_t> visiting_;
  std::vector<Walk> visited_;
  std::vector<uint32_t> current_;
  mutable std::vector<uint32_t> tmp_;
  unsigned setback_ = 0;
  uint64_t entropy_ = 1;
};

// get_uniq_full_haplotype with tries = 0: walk to a sink, keep the walk
std::vector<uint32_t> full_walk(const Graph& g, Haplotyper& hp)
{
  std::vector<uint32_t> walk;
  const uint64_t cap = 4 * g.n_nodes() + 16;       // a cyclic graph has no sink to stop at
  while (hp.value() != END && walk.size() < cap) {
    walk.push_back(hp.value());
    hp.next();
  }
  return walk;
}

struct Patch { size_t first, last; uint32_t head, tail; };      // nodes walk[first..last], trimmed

// maximal runs of `context`-base windows of `walk` whose node sequence none of `earlier` contains
std::vector<Patch> cut_patches(const Graph& g, const std::vector<uint32_t>& walk,
                               const std::vector<std::vector<uint32_t>>& earlier, uint32_t context,
                               std::vector<uint32_t>& at /* one entry per node, all NO_NODE; left so */)
{
  const size_t m = walk.size();
  std::vector<Patch> out;
  if (m == 0) return out;
  std::vector<uint64_t> pos;                       // first base of walk[i] in the walk's sequence
  resize_populated(pos, m + 1);
  for (size_t i = 0; i < m; ++i) pos[i + 1] = pos[i] + g.node_len(walk[i]);
  const uint64_t L = pos[m];
  if (L == 0) return out;
  // run[i]: the longest j - i + 1 such that walk[i..j] is a contiguous run of some earlier walk
  std::vector<uint32_t> run, cur;
  resize_populated(run, m); resize_populated(cur, m);
  const bool sequential = getenv("PSIGPU_TEST_SEQ_PATCHES") != nullptr;      // tests: the one-thread formulation
  for (const auto& V : earlier) {
    // node -> its (first) position in V: a plain array over the nodes (a hash map of a whole-genome walk's
    // 300 M nodes took longer than everything else in the index build)
    const int64_t nv = (int64_t)V.size(), mm = (int64_t)m;
    bool increasing = !sequential;
    if (increasing) {
#pragma omp parallel for reduction(&& : increasing)
      for (int64_t q = 1; q < nv; ++q) increasing = increasing && V[q - 1] < V[q];
    }
    if (!increasing) {
      for (size_t q = V.size(); q-- > 0;) at[V[q]] = (uint32_t)q;
      for (size_t i = m; i-- > 0;) {
        const uint32_t q = at[walk[i]];
        if (q == NO_NODE) { cur[i] = 0; continue; }
        cur[i] = (i + 1 < m && (size_t)q + 1 < V.size() && V[q + 1] == walk[i + 1] && cur[i + 1]) ? cur[i + 1] + 1 : 1;
        run[i] = std::max(run[i], cur[i]);
      }
      for (uint32_t v : V) at[v] = NO_NODE;
      continue;
    }
    // The same in parallel (V visits no node twice).  cur[i] = 1 + (link(i) ? cur[i + 1] : 0) for a node of V, where
    // link(i) -- walk[i + 1] follows walk[i] in V -- depends on V and the walk alone: a run is a chain of links that
    // ends in a node without one, and cur[i] is the distance to that node.  State per node (0 not in V, 1 no link,
    // 2 link), then a backward pass per chunk that carries the position of the next state-1 node, seeded with the
    // first such node behind the chunk.
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nv; ++q) at[V[q]] = (uint32_t)q;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < mm; ++i) {
      const uint32_t q = at[walk[i]];
      cur[i] = q == NO_NODE ? 0u : ((i + 1 < mm && (int64_t)q + 1 < nv && V[q + 1] == walk[i + 1]) ? 2u : 1u);
    }
    const int64_t CH = 1 << 20, n_ch = (mm + CH - 1) / CH;
    std::vector<int64_t> next_one(n_ch + 1, mm);        // first state-1 node at or behind the chunk's start
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n_ch; ++c) {
      const int64_t a = c * CH, b = std::min(mm, a + CH);
      int64_t f = mm;
      for (int64_t i = a; i < b; ++i) if (cur[i] == 1u) { f = i; break; }
      next_one[c] = f;
    }
    for (int64_t c = n_ch - 1; c >= 0; --c) if (next_one[c] == mm) next_one[c] = next_one[c + 1];
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < n_ch; ++c) {
      const int64_t a = c * CH, b = std::min(mm, a + CH);
      int64_t carry = next_one[c + 1];
      for (int64_t i = b; i-- > a;) {
        const uint32_t st = cur[i];
        if (st == 1u) carry = i;
        const uint32_t v = st ? (uint32_t)(carry - i + 1) : 0u;
        cur[i] = v;
        if (v > run[i]) run[i] = v;
      }
    }
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nv; ++q) at[V[q]] = NO_NODE;
  }
  // Windows start at x in [0, x_max]; the one starting in node i at x ends in the node holding base
  // x + c - 1, and is uncovered iff that node lies behind walk[i + run[i] - 1]
  const uint64_t c = std::min<uint64_t>(context, L);
  const uint64_t x_max = L - c;
  uint64_t open_s = 0, open_t = 0;                 // the run of uncovered windows being collected: bases [s, t)
  size_t open_i = 0, last_i = 0;                   // the node it was opened in / last extended in
  bool open = false;
  auto close = [&]() {
    if (!open) return;
    // the nodes holding the first and the last base of the run: the node the run was opened in, and a few nodes
    // behind the one it was last extended in (two bisections of a 200 M-entry array per patch, 26 M patches,
    // were half of the cutter's time)
    size_t a = open_i, b = last_i;
    while (b + 1 < m && pos[b + 1] <= open_t - 1) ++b;
    if (sequential) {
      a = std::upper_bound(pos.begin(), pos.end(), open_s) - pos.begin() - 1;
      b = std::upper_bound(pos.begin(), pos.end(), open_t - 1) - pos.begin() - 1;
    }
    while (a < b && g.node_len(walk[a]) == 0) ++a;
    Patch p{ a, b, (uint32_t)(open_s - pos[a]), (uint32_t)(open_t - pos[b]) };
    if (p.tail == g.node_len(walk[b])) p.tail = 0;
    out.push_back(p);
    open = false;
  };
  for (size_t i = 0; i < m; ++i) {
    if (pos[i] > x_max) break;
    if (pos[i + 1] == pos[i]) continue;
    const uint64_t covered_to = run[i] ? pos[i + run[i]] : pos[i];     // first base behind the covered run
    // uncovered windows starting in this node: x + c - 1 >= covered_to
    uint64_t lo = covered_to >= c ? covered_to - c + 1 : 0;
    lo = std::max<uint64_t>(lo, pos[i]);
    const uint64_t hi = std::min<uint64_t>(pos[i + 1] - 1, x_max);      // last window start in this node
    if (lo > hi) continue;
    if (open && lo > open_t) close();             // a gap of covered bases: the run ended
    if (!open) { open = true; open_s = lo; open_i = i; }
    open_t = hi + c;
    last_i = i;
  }
  close();
  return out;
}

}  // namespace

void pick_paths(const Graph& g, uint32_t n_per_region, bool patched, uint32_t context, uint64_t rng_seed,
                std::vector<std::vector<uint32_t>>& out, std::vector<uint32_t>& head, std::vector<uint32_t>& tail)
{
  out.clear(); head.clear(); tail.clear();
  if (n_per_region == 0) return;
  Haplotyper hp(g, rng_seed);
  std::vector<uint32_t> at;                        // cut_patches' node -> position array, made on first use
  for (size_t r = 0; r < g.paths.size(); ++r) {
    if (g.paths[r].empty()) continue;
    hp.reset(g.paths[r][0]);                       // the region's walks start where its embedded path starts (:1159-1160)
    std::vector<std::vector<uint32_t>> walks;      // this region's walks so far (whole, also when patches are kept)
    if (patched && n_per_region > 1 && at.empty()) at.assign(g.n_nodes(), NO_NODE);
    for (uint32_t i = 0; i < n_per_region; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<uint32_t> walk = full_walk(g, hp);
      const auto t1 = std::chrono::steady_clock::now();
      hp.save();
      if (getenv("PSIGPU_TRACE_PICK")) fprintf(stderr, "[psigpu]   walk %u: %zu nodes, walk %.2f s, save %.2f s\n", i, walk.size(),
                                               std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
      if (walk.empty()) continue;
      if (!patched || walks.empty()) {             // get_uniq_patched_haplotype: level 0 gives a full haplotype (:568-571)
        out.push_back(walk); head.push_back(0); tail.push_back(0);
      } else {
        const auto t2 = std::chrono::steady_clock::now();
        const std::vector<Patch> pt = cut_patches(g, walk, walks, context, at);
        const auto t3 = std::chrono::steady_clock::now();
        out.reserve(out.size() + pt.size()); head.reserve(head.size() + pt.size()); tail.reserve(tail.size() + pt.size());
        for (const Patch& p : pt) {
          out.emplace_back(walk.begin() + p.first, walk.begin() + p.last + 1);
          head.push_back(p.head); tail.push_back(p.tail);
        }
        if (getenv("PSIGPU_TRACE_PICK")) fprintf(stderr, "[psigpu]   patches %zu: cut %.2f s, copy %.2f s\n", pt.size(),
                                                 std::chrono::duration<double>(t3 - t2).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t3).count());
      }
      walks.push_back(std::move(walk));
    }
  }
}

}  // namespace psigpu
