// Path selection, FM-index construction, starting-loci detection, (de)serialisation.
//
// Host-side and one-off; corresponds to SeedFinder::create_path_index
// (reference include/psi/seed_finder.hpp:1330-1355) restricted to full (un-patched) paths:
//   pick_paths           :1138-1167  ->  pick_paths()
//   index_paths          :1169-1176  ->  build_index()  (own SA-IS instead of sdsl::construct,
//                                        include/psi/fmindex.hpp:257-271)
//   add_uncovered_loci   :1481-1541  ->  find_starting_loci()
//   add_all_loci         :1543-1585  ->  find_starting_loci() with no paths
// The data layout produced here is this library's own; results are compared
// with the reference as hit SETS, never as suffix-array coordinates.
#include <algorithm>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sys/stat.h>
#include <random>

#include "host.hpp"
#include "sais.hpp"

namespace psigpu {

// ------------------------------------------------------------------------------------
// Starting loci: a locus (v, o) is a starting locus iff at least one k-walk from it is not spelled
// by a contiguous run of an indexed path (seed_finder.hpp:1481-1541, step 1) -- "spelled" in the
// indexed TEXT: a patched path (pathindex.hpp:496-560) covers its first node only from its head
// offset on and its last node only up to its tail length, so a walk that needs the bases cut away
// is not covered by it.  (The reference asks covered_by() about node ids alone and relies on the
// patching rules for the bases; the definition here is the one that keeps the hit set complete.)
//
// Coverage is tracked with path STEPS: the paths' node lists are concatenated, a step is an index
// into that array.  A walk w0 .. wm is covered iff some step s has node[s .. s+m] = w0 .. wm inside
// one path, the walk's first base is at or behind that step's first indexed base and its last base
// in front of step s+m's end.  The set of candidate steps only shrinks while a walk is extended and
// holds at most as many steps as paths pass through the node, whatever the number of paths:
// thousands of patches cost no more than the haplotypes they were cut from.
// ------------------------------------------------------------------------------------
namespace {

struct LociCtx {
  const Graph& g;
  uint32_t k;
  std::vector<uint32_t> step_node;              // concatenated path node lists
  std::vector<uint32_t> step_lo, step_hi;       // indexed bases [lo, hi) of the step's node
  std::vector<uint8_t> step_last;               // last step of its path
  std::vector<uint64_t> at_off;                 // CSR: steps at every node
  std::vector<uint32_t> at;
  std::vector<uint32_t> reach;      // max bases spelled by a walk starting at node start, capped at k
  std::vector<uint32_t> child;      // max reach over the out-neighbours (0 for sinks)
};

inline void mark(uint64_t* unc, uint32_t lo, uint32_t hi)      // bits lo .. hi (need lengths, < 64)
{
  for (uint32_t i = lo; i <= hi && i < 64; ++i) *unc |= 1ull << i;
}

// The walk has spelled S bases behind its first node and now enters u; `state` = the steps the walk is
// still a run of, standing on the node before u.  Marks in `unc` (bit i = need i) every extension
// length for which an uncovered walk exists.
void explore(const LociCtx& c, uint32_t u, uint32_t S, const uint32_t* state, size_t n_state, uint64_t* unc, uint32_t depth)
{
  if (depth > 4 * c.k) return;      // guards cycles of empty nodes
  const uint32_t len = (uint32_t)c.g.node_len(u);
  // the candidate steps that go on through u: at most as many as there were, nearly always a handful -- kept on the
  // stack (a heap vector per call made 128 threads queue at the allocator: 4.4 s for 300 M nodes)
  uint32_t small[32];
  std::vector<uint32_t> big;
  uint32_t* next = small;
  if (n_state > 32) { big.resize(n_state); next = big.data(); }
  size_t n_next = 0;
  uint32_t hi_max = 0;              // most bases of u any candidate path still spells
  for (size_t i = 0; i < n_state; ++i) {
    const uint32_t t = state[i] + 1;       // (a candidate is never the last step of its path)
    if (c.step_node[t] != u) continue;
    hi_max = std::max(hi_max, c.step_hi[t]);
    if (!c.step_last[t]) next[n_next++] = t;
  }
  const uint32_t kmax = c.k - 1;    // needs are 1 .. k-1
  // needs that end inside u: covered while some candidate spells that many of u's bases
  if (len && S + hi_max < kmax && hi_max < len) mark(unc, S + hi_max + 1, std::min(kmax, S + len));
  const uint32_t S2 = S + len;
  if (S2 >= kmax) return;
  if (n_next == 0) {                // every longer walk through u is uncovered
    if (c.child[u]) mark(unc, S2 + 1, std::min(kmax, S2 + c.child[u]));
    return;
  }
  for (uint64_t e = c.g.edge_off[u]; e < c.g.edge_off[u + 1]; ++e) explore(c, c.g.edge_to[e], S2, next, n_next, unc, depth + 1);
}

}  // namespace

void find_starting_loci(const Graph& g, const std::vector<std::vector<uint32_t>>& paths,
                        const std::vector<uint32_t>& path_head, const std::vector<uint32_t>& path_tail,
                        uint32_t k, uint32_t step, std::vector<uint32_t>& loci_node,
                        std::vector<uint32_t>& loci_off)
{
  loci_node.clear();
  loci_off.clear();
  if (step == 0) step = 1;
  const uint64_t n = g.n_nodes();
  LociCtx c{ g, k, {}, {}, {}, {}, {}, {}, {}, {} };
  const bool trace = getenv("PSIGPU_TRACE") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[psigpu]   loci: %s %.2f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  resize_populated(c.reach, n);
  resize_populated(c.child, n);
  // reach: fixed point of reach(u) = min(k, len(u) + max_child reach(child)); len 0 nodes allowed
  for (uint64_t v = 0; v < n; ++v) c.reach[v] = (uint32_t)std::min<uint64_t>(k, g.node_len((uint32_t)v));
  // (sweeps from the last node to the first, chunks of nodes in parallel: a chunk may read a neighbour's value of
  // the sweep before -- values only grow towards the one least fixed point, and the loop ends with a sweep in which
  // nothing changed, whose `child` values are therefore final.  A walk of k bases reaches a handful of nodes ahead,
  // so stale values matter at chunk borders only: two or three sweeps.)
  bool changed = true;
  const int64_t RCH = 1 << 18, n_rch = ((int64_t)n + RCH - 1) / RCH;
  while (changed) {
    changed = false;
#pragma omp parallel for schedule(dynamic, 1) reduction(|| : changed)
    for (int64_t ch = n_rch - 1; ch >= 0; --ch) {
      const uint64_t v0 = (uint64_t)ch * RCH, v1 = std::min<uint64_t>(n, v0 + RCH);
      for (uint64_t v = v1; v-- > v0;) {
        uint32_t best = 0;
        for (uint64_t e = g.edge_off[v]; e < g.edge_off[v + 1]; ++e)
          best = std::max(best, __atomic_load_n(&c.reach[g.edge_to[e]], __ATOMIC_RELAXED));
        uint32_t r = (uint32_t)std::min<uint64_t>(k, g.node_len((uint32_t)v) + best);
        c.child[v] = best;
        if (r > c.reach[v]) { __atomic_store_n(&c.reach[v], r, __ATOMIC_RELAXED); changed = true; }
      }
    }
  }
  lap("reach");
  // path steps and the steps at every node (all loops over the paths / steps / nodes in parallel: 600 M steps at
  // whole-genome size; a node's steps are sorted so that the lists are what one sequential pass would make)
  {
    const int64_t np = (int64_t)paths.size();
    std::vector<uint64_t> first(paths.size() + 1, 0);            // first step of every path
    for (size_t p = 0; p < paths.size(); ++p) first[p + 1] = first[p] + paths[p].size();
    const uint64_t total = first[paths.size()];
    resize_populated(c.step_node, total); resize_populated(c.step_lo, total); resize_populated(c.step_hi, total);
    resize_populated(c.step_last, total);
    resize_populated(c.at_off, n + 1);
    // work items: runs of short paths (patches), slices of long ones (a whole-genome walk is one path of 200 M steps)
    struct StepItem { size_t pa, pb, i0, i1; };
    std::vector<StepItem> items;
    {
      const size_t TARGET = 1u << 20;
      size_t ga = 0, gsteps = 0;
      for (size_t p = 0; p < paths.size(); ++p) {
        const size_t m = paths[p].size();
        if (m > TARGET) {
          if (p > ga) items.push_back({ ga, p, 0, 0 });
          for (size_t a0 = 0; a0 < m; a0 += TARGET) items.push_back({ p, p + 1, a0, std::min(m, a0 + TARGET) });
          ga = p + 1; gsteps = 0;
        } else if ((gsteps += m) >= TARGET) { items.push_back({ ga, p + 1, 0, 0 }); ga = p + 1; gsteps = 0; }
      }
      if (paths.size() > ga) items.push_back({ ga, paths.size(), 0, 0 });
    }
    (void)np;
    const int64_t n_items = (int64_t)items.size();
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t it = 0; it < n_items; ++it) {
      const StepItem w = items[it];
      for (size_t p = w.pa; p < w.pb; ++p) {
        const auto& P = paths[p];
        const size_t i0 = w.i1 ? w.i0 : 0, i1 = w.i1 ? w.i1 : P.size();
        uint64_t s = first[p] + i0;
        for (size_t i = i0; i < i1; ++i, ++s) {
          const uint32_t len = (uint32_t)g.node_len(P[i]);
          uint32_t lo = 0, hi = len;
          if (i == 0 && p < path_head.size()) lo = std::min(len, path_head[p]);
          if (i + 1 == P.size() && p < path_tail.size() && path_tail[p]) hi = std::min(len, path_tail[p]);
          c.step_node[s] = P[i]; c.step_lo[s] = lo; c.step_hi[s] = hi;
          c.step_last[s] = i + 1 == P.size();
          __atomic_fetch_add(&c.at_off[P[i] + 1], 1ull, __ATOMIC_RELAXED);
        }
      }
    }
    for (uint64_t v = 0; v < n; ++v) c.at_off[v + 1] += c.at_off[v];
    resize_populated(c.at, total);
    std::vector<uint64_t> fill;
    resize_populated(fill, n);
    {
      const int64_t nn2 = (int64_t)n;
#pragma omp parallel for schedule(static)
      for (int64_t v = 0; v < nn2; ++v) fill[v] = c.at_off[v];
    }
    const int64_t nt = (int64_t)total;
#pragma omp parallel for schedule(static)
    for (int64_t s = 0; s < nt; ++s) c.at[__atomic_fetch_add(&fill[c.step_node[s]], 1ull, __ATOMIC_RELAXED)] = (uint32_t)s;
    const int64_t nn = (int64_t)n;
#pragma omp parallel for schedule(dynamic, 1 << 16)
    for (int64_t v = 0; v < nn; ++v)
      if (c.at_off[v + 1] - c.at_off[v] > 1) std::sort(c.at.begin() + c.at_off[v], c.at.begin() + c.at_off[v + 1]);
  }
  lap("path steps");
  // Nodes are independent: blocks of nodes in parallel (OpenMP), each block's loci in node order,
  // blocks concatenated in order.
  const uint64_t BLK = 1u << 16;
  const uint64_t n_blk = (n + BLK - 1) / BLK;
  std::vector<std::vector<uint32_t>> bn(n_blk), bo(n_blk);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t b = 0; b < (int64_t)n_blk; ++b) {
    std::vector<uint32_t>& out_n = bn[b];
    std::vector<uint32_t>& out_o = bo[b];
    std::vector<uint32_t> state, heads;
    std::vector<uint64_t> unc_of;
    const uint64_t v1 = std::min<uint64_t>(n, (uint64_t)(b + 1) * BLK);
    for (uint64_t v = (uint64_t)b * BLK; v < v1; ++v) {
      const uint32_t len = (uint32_t)g.node_len((uint32_t)v);
      if (len == 0) continue;
      const uint32_t* sv = c.at.data() + c.at_off[v];
      const uint32_t nsv = (uint32_t)(c.at_off[v + 1] - c.at_off[v]);
      // The steps a walk from offset o can be a run of are those whose first indexed base is at or
      // before o: the distinct head offsets cut the node into ranges with one candidate set each.
      heads.clear();
      for (uint32_t i = 0; i < nsv; ++i) heads.push_back(c.step_lo[sv[i]]);
      std::sort(heads.begin(), heads.end());
      heads.erase(std::unique(heads.begin(), heads.end()), heads.end());
      unc_of.assign(heads.size(), 0);
      for (size_t h = 0; h < heads.size(); ++h) {
        // walks that leave v: candidates spell v to its end and go on
        state.clear();
        for (uint32_t i = 0; i < nsv; ++i) {
          const uint32_t s = sv[i];
          if (c.step_lo[s] <= heads[h] && c.step_hi[s] == len && !c.step_last[s]) state.push_back(s);
        }
        uint64_t unc = 0;
        if (state.empty()) { if (c.child[v]) mark(&unc, 1, std::min(k - 1, c.child[v])); }
        else
          for (uint64_t e = g.edge_off[v]; e < g.edge_off[v + 1]; ++e) explore(c, g.edge_to[e], 0, state.data(), state.size(), &unc, 0);
        unc_of[h] = unc;
      }
      uint32_t since = 0;            // locus subsampling (psikt -e): every step-th starting locus per node
      for (uint32_t o = 0; o < len; ++o) {
        if ((uint64_t)len - o + c.child[v] < k) continue;      // no k-walk starts here
        bool take;
        const int64_t need = (int64_t)k - (int64_t)(len - o);
        if (need <= 0) {
          // the k-walk lies inside v: covered iff one step spells [o, o + k)
          take = true;
          for (uint32_t i = 0; i < nsv && take; ++i)
            if (c.step_lo[sv[i]] <= o && c.step_hi[sv[i]] >= o + k) take = false;
        } else {
          // the range of o: the last head offset <= o (none: no candidate at all)
          size_t h = std::upper_bound(heads.begin(), heads.end(), o) - heads.begin();
          take = h == 0 ? true : ((unc_of[h - 1] >> need) & 1) != 0;
        }
        if (!take) continue;
        if (since % step == 0) {
          out_n.push_back((uint32_t)v);
          out_o.push_back(o);
        }
        ++since;
      }
    }
  }
  lap("nodes");
  std::vector<uint64_t> at_blk(n_blk + 1, 0);
  for (uint64_t b = 0; b < n_blk; ++b) at_blk[b + 1] = at_blk[b] + bn[b].size();
  const uint64_t total = at_blk[n_blk];
  resize_populated(loci_node, total);
  resize_populated(loci_off, total);
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t b = 0; b < (int64_t)n_blk; ++b) {
    std::copy(bn[b].begin(), bn[b].end(), loci_node.begin() + at_blk[b]);
    std::copy(bo[b].begin(), bo[b].end(), loci_off.begin() + at_blk[b]);
    std::vector<uint32_t>().swap(bn[b]);
    std::vector<uint32_t>().swap(bo[b]);
  }
  lap("gather");
}

// ------------------------------------------------------------------------------------
// FM-index over the forward concatenation of the path sequences.
// ------------------------------------------------------------------------------------
static inline int base_sym(char ch)
{
  switch (ch) {
    case 'A': case 'a': return SYM_A;
    case 'C': case 'c': return SYM_C;
    case 'G': case 'g': return SYM_G;
    case 'T': case 't': return SYM_T;
    default: return -1;
  }
}

// FM arrays + segment table over the concatenation of paths [p0, p1): one PART of the index (an index
// is one part unless its text would pass the 32-bit row limit).  `head` / `tail`: per-path trimming.
static int build_part(const Graph& g, const psigpu_index_opts& opts, uint32_t sa_rate, bool keep,
                      const std::vector<std::vector<uint32_t>>& paths, const std::vector<uint32_t>& head,
                      const std::vector<uint32_t>& tail, size_t p0, size_t p1, Index* x, std::string* err)
{
  x->sa_rate = sa_rate;
  // ---- text + segments -----------------------------------------------------------
  // The concatenation is assembled by work items of ~1 M path steps (several short paths -- patches -- or a slice
  // of a long one): a counting pass, a prefix sum, a filling pass, both passes over the items in parallel.  What
  // an item has to know of the text in front of it is one bit -- was the last symbol the separator of an N run
  // (in_gap) -- which it reads off the steps before its first one.  (One sequential pass of push_backs took 25 of
  // the 75 s of a whole-genome build.)
  std::vector<uint8_t> T;
  const auto t_text = std::chrono::steady_clock::now();
  auto& ss = x->seg_start; auto& sn = x->seg_node; auto& so = x->seg_noff;
  struct Item { size_t pa, pb; size_t s0, s1; bool in_gap; uint64_t n_sym, n_seg; };      // paths [pa, pb); a slice: pb == pa + 1, steps [s0, s1), s1 != 0
  std::vector<Item> items;
  std::vector<uint8_t> sep_before(p1 - p0, 0);    // a non-empty path behind another non-empty one starts with a separator
  {
    size_t TARGET = 1u << 20;
    if (const char* e = getenv("PSIGPU_TEST_TEXT_ITEM")) TARGET = std::max<size_t>(1, strtoul(e, nullptr, 10));      // tests: items of a few steps
    bool any_before = false;
    size_t ga = p0, gsteps = 0;
    auto flush_group = [&](size_t pb) {
      if (pb > ga) items.push_back(Item{ ga, pb, 0, 0, false, 0, 0 });
      ga = pb; gsteps = 0;
    };
    for (size_t pi = p0; pi < p1; ++pi) {
      const size_t m = paths[pi].size();
      sep_before[pi - p0] = m && any_before;
      if (m) any_before = true;
      if (m > TARGET) {
        flush_group(pi);
        for (size_t a0 = 0; a0 < m; a0 += TARGET) items.push_back(Item{ pi, pi + 1, a0, std::min(m, a0 + TARGET), false, 0, 0 });
        ga = pi + 1;
      } else {
        gsteps += m;
        if (gsteps >= TARGET) flush_group(pi + 1);
      }
    }
    flush_group(p1);
  }
  // the bases [o_begin, len) a step contributes (a patched path starts at its head offset and ends after its
  // tail length: Path::left / right, path_base.hpp:113-114, :240-246)
  auto step_range = [&](size_t pi, size_t si, uint64_t* o_begin, uint64_t* len) {
    const auto& P = paths[pi];
    *o_begin = si == 0 ? head[pi] : 0;
    *len = (si + 1 == P.size() && tail[pi]) ? tail[pi] : g.node_len(P[si]);
  };
  for (Item& it : items)
    if (it.s0) {                                  // a slice behind the first: the state the steps before it leave
      for (size_t si = it.s0; si-- > 0;) {
        uint64_t ob, ln;
        step_range(it.pa, si, &ob, &ln);
        if (ob < ln) { it.in_gap = base_sym(g.labels[g.label_off[paths[it.pa][si]] + ln - 1]) < 0; break; }
      }
    }
  // one item: FILL = false counts symbols and segments, FILL = true writes them at the item's offsets
  auto run_item = [&](Item& it, bool fill, uint64_t t_at, uint64_t s_at) {
    uint64_t nt = 0, ns = 0;
    for (size_t pi = it.pa; pi < it.pb; ++pi) {
      const auto& P = paths[pi];
      if (P.empty()) continue;
      const bool slice = it.s1 != 0;
      const size_t sa = slice ? it.s0 : 0, sb = slice ? it.s1 : P.size();
      bool in_gap = false;                        // last emitted symbol was a separator for an N run
      if (sa == 0) {
        if (sep_before[pi - p0]) { if (fill) T[t_at + nt] = SYM_SEP; ++nt; }
      } else in_gap = it.in_gap;
      for (size_t si = sa; si < sb; ++si) {
        const uint32_t v = P[si];
        const char* lab = g.labels.data() + g.label_off[v];
        uint64_t o_begin, len;
        step_range(pi, si, &o_begin, &len);
        bool open = false;                        // a segment of this node is open
        for (uint64_t o = o_begin; o < len; ++o) {
          const int sy = base_sym(lab[o]);
          if (sy < 0) {
            if (!in_gap) { if (fill) T[t_at + nt] = SYM_SEP; ++nt; in_gap = true; }
            open = false;
            continue;
          }
          if (!open) {
            if (fill) { ss[s_at + ns] = (uint32_t)(t_at + nt); sn[s_at + ns] = v; so[s_at + ns] = (uint32_t)o; }
            ++ns;
            open = true;
          }
          in_gap = false;
          if (fill) T[t_at + nt] = (uint8_t)sy;
          ++nt;
        }
      }
    }
    it.n_sym = nt; it.n_seg = ns;
  };
  const int64_t n_items = (int64_t)items.size();
  const auto t_a = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t i = 0; i < n_items; ++i) run_item(items[i], false, 0, 0);
  const auto t_b = std::chrono::steady_clock::now();
  if (getenv("PSIGPU_TRACE")) fprintf(stderr, "[psigpu]   omp threads %d; items %.2f s, count pass %.2f s\n", omp_get_max_threads(), std::chrono::duration<double>(t_a - t_text).count(), std::chrono::duration<double>(t_b - t_a).count());
  {
    uint64_t tt = 0, tsg = 0;
    std::vector<uint64_t> t_at(items.size()), s_at(items.size());
    for (size_t i = 0; i < items.size(); ++i) { t_at[i] = tt; s_at[i] = tsg; tt += items[i].n_sym; tsg += items[i].n_seg; }
    if (tt + 1 >= (1ull << 32)) { *err = "index part too large (text beyond 2^32 symbols)"; return PSIGPU_ERR_ARG; }
    T.reserve(tt + 1);
    ss.reserve(tsg + 2); sn.reserve(tsg + 1); so.reserve(tsg + 1);
    // (a dozen GB at whole-genome size: the pages are faulted in by all threads before resize() zeroes them on one)
    populate_pages(T.data(), tt + 1); populate_pages(ss.data(), (tsg + 2) * 4); populate_pages(sn.data(), (tsg + 1) * 4); populate_pages(so.data(), (tsg + 1) * 4);
    T.resize(tt);
    ss.resize(tsg); sn.resize(tsg); so.resize(tsg);
    const auto t_c = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < n_items; ++i) run_item(items[i], true, t_at[i], s_at[i]);
    if (getenv("PSIGPU_TRACE")) fprintf(stderr, "[psigpu]   resize %.2f s, fill pass %.2f s\n", std::chrono::duration<double>(t_c - t_b).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_c).count());
  }
  T.push_back(SYM_END);
  const uint64_t n = T.size();
  if (getenv("PSIGPU_TRACE"))
    fprintf(stderr, "[psigpu]   part text: %llu symbols, %zu segments, %zu work items, %.2f s\n", (unsigned long long)n, ss.size(), items.size(),
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_text).count());
  x->n = n;
  x->fm_ok = true;
  x->exc_shift = EXC_SUPER_SHIFT;
  if (const char* e = getenv("PSIGPU_TEST_EXC_SHIFT")) x->exc_shift = std::min<uint32_t>(EXC_SUPER_SHIFT, (uint32_t)strtoul(e, nullptr, 10));   // tests: tiny super-blocks
  if (ss.empty() || ss[0] != 0) {          // position 0 must belong to a segment
    ss.insert(ss.begin(), 0); sn.insert(sn.begin(), NO_NODE); so.insert(so.begin(), 0);
  }
  ss.push_back((uint32_t)n);
  {
    uint64_t nd = (n >> DIR_SHIFT) + 1;
    x->seg_dir.resize(nd);
    uint64_t s = 0;
    for (uint64_t i = 0; i < nd; ++i) {
      uint64_t pos = i << DIR_SHIFT;
      while (s + 1 < sn.size() && ss[s + 1] <= pos) ++s;
      x->seg_dir[i] = (uint32_t)s;
    }
  }

  // interval-table length
  uint32_t q = opts.ftab_len;
  if (q == 0) {                       // auto: ceil(log4 n); at most 13 (512 MiB) when built on the host,
    q = 1;                            // 15 (8 GiB: whole-genome texts) when built on the device -- 16 can be asked for, but the
                                      // 32-GiB table was measured slower at 2.95 G symbols (its own misses cost more than the rows it saves)
    const uint32_t q_max = opts.build_on_device ? 15 : 13;
    while (q < q_max && (1ull << (2 * q)) < n) ++q;
  }
  if (q == 0xFFFFFFFFu || p0 == p1) q = 0;
  // (the host builder marks "no q-mer here" with a 32-bit all-ones code: 16-mers need the device builder)
  if (q > 16 || (q == 16 && !opts.build_on_device)) {
    *err = "ftab_len above 16 (15 for host builds)"; return PSIGPU_ERR_ARG;
  }

  std::vector<int32_t> SA;
  if (opts.build_on_device) {
    // suffix array, rank blocks, samples, exceptions, interval table, 4-bit text on the GPU
    int st = gpu_build_fm(T, sa_rate, q, (int)opts.build_on_device - 1, x, keep ? &SA : nullptr, err);
    if (st != PSIGPU_OK) return st;
  } else {
  // ---- suffix array ----------------------------------------------------------------
  SA.resize(n);
  suffix_array(T.data(), SA.data(), (int32_t)n, 6);

  // ---- BWT rank blocks, samples, exceptions -----------------------------------------
  uint64_t nblk = n / BLOCK_SYMS + 1;
  x->blocks.assign(nblk, RankBlock{ { 0, 0, 0 }, 0, { 0, 0, 0, 0, 0, 0 } });
  x->samples.resize((n + sa_rate - 1) / sa_rate);
  uint64_t cnt[4] = { 0, 0, 0, 0 }, nexc = 0, nsep = 0;
  const uint32_t xs = x->exc_shift;
  x->exc_super.assign(((nblk - 1) >> xs) + 1, 0);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t b = i / BLOCK_SYMS, j = i % BLOCK_SYMS;
    if (j == 0) {
      RankBlock& B = x->blocks[b];
      B.cnt[0] = (uint32_t)cnt[0]; B.cnt[1] = (uint32_t)cnt[1]; B.cnt[2] = (uint32_t)cnt[2];
      if ((b & ((1ull << xs) - 1)) == 0) x->exc_super[b >> xs] = (uint32_t)nexc;
      B.exc = (uint32_t)((nexc - x->exc_super[b >> xs]) << 8);
    }
    uint8_t c = SA[i] ? T[SA[i] - 1] : T[n - 1];
    uint64_t two;
    if (c >= SYM_A) { two = c - SYM_A; ++cnt[two]; }
    else {
      two = 0;
      x->exc_row.push_back((uint32_t)i);
      x->exc_sa.push_back((uint32_t)SA[i]);
      ++nexc;
      RankBlock& B = x->blocks[b];
      if ((B.exc & 0xFF) < 255) ++B.exc;
      if (c == SYM_SEP) ++nsep;
    }
    // bit planes per group of 64 symbols: word 2g = low bits, word 2g+1 = high bits
    x->blocks[b].sym[2 * (j >> 6)] |= (two & 1) << (j & 63);
    x->blocks[b].sym[2 * (j >> 6) + 1] |= (two >> 1) << (j & 63);
    if (i % sa_rate == 0) x->samples[i / sa_rate] = (uint32_t)SA[i];
  }
  if (n % BLOCK_SYMS == 0) {
    RankBlock& B = x->blocks[nblk - 1];
    B.cnt[0] = (uint32_t)cnt[0]; B.cnt[1] = (uint32_t)cnt[1]; B.cnt[2] = (uint32_t)cnt[2];
    if (((nblk - 1) & ((1ull << xs) - 1)) == 0) x->exc_super[(nblk - 1) >> xs] = (uint32_t)nexc;
    B.exc = (uint32_t)((nexc - x->exc_super[(nblk - 1) >> xs]) << 8);
  }
  x->C[0] = 1 + nsep;
  x->C[1] = x->C[0] + cnt[0];
  x->C[2] = x->C[1] + cnt[1];
  x->C[3] = x->C[2] + cnt[2];

  // ---- the text itself, 4 bits per symbol ---------------------------------------------------
  x->text4.assign(n / 16 + 2, 0);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t nib = T[i] >= SYM_A ? (uint64_t)(T[i] - SYM_A) : 4ull;
    x->text4[i >> 4] |= nib << (60 - 4 * (i & 15));
  }

  // ---- interval table for the last q bases of a seed -----------------------------------
  {
    x->ftab_len = q;
    if (q) {
      // code[p] = 2-bit code of T[p, p+q) (first base most significant) or NONE
      const uint32_t NONE = 0xFFFFFFFFu;
      std::vector<uint32_t> code(n, NONE);
      uint32_t mask = (uint32_t)((1ull << (2 * q)) - 1), run = 0, good = 0;
      for (uint64_t i = 0; i < n; ++i) {          // rolling over windows ending at i
        uint8_t c = T[i];
        if (c >= SYM_A) { run = ((run << 2) | (uint32_t)(c - SYM_A)) & mask; ++good; }
        else { run = 0; good = 0; }
        if (good >= q) code[i + 1 - q] = run;
      }
      x->ftab.assign(2ull << (2 * q), 0);
      for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = code[SA[i]];
        if (c == NONE) continue;
        uint32_t* e = &x->ftab[2ull * c];
        if (e[1] == 0) e[0] = (uint32_t)i;
        e[1] = (uint32_t)i + 1;
      }
    }
  }

  }

  if (keep) { x->text = std::move(T); x->sa = std::move(SA); }
  return PSIGPU_OK;
}

uint64_t graph_fingerprint(const Graph& g)
{
  // node count, total label length, edge count, node ids, label offsets, edge targets and the label bytes:
  // enough to tell that an index file was made for another graph -- other sequences or other edges under
  // the same ids included (load_path_index then rebuilds, as the reference does when its loci file does
  // not match: seed_finder.hpp:1396-1413).  Blocks are hashed in parallel and combined in order.
  auto mix = [](uint64_t h, uint64_t x) { h ^= x; h *= 0x100000001b3ull; h ^= h >> 29; return h; };
  uint64_t h = 0xcbf29ce484222325ull;
  h = mix(h, g.n_nodes()); h = mix(h, g.labels.size()); h = mix(h, g.edge_to.size());
  auto hash_range = [&](auto get, uint64_t n) {
    const uint64_t BLK = 1u << 20, nb = (n + BLK - 1) / BLK;
    std::vector<uint64_t> part(nb);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)nb; ++b) {
      uint64_t x = 0x9e3779b97f4a7c15ull ^ (uint64_t)b;
      const uint64_t e = std::min<uint64_t>(n, ((uint64_t)b + 1) * BLK);
      for (uint64_t i = (uint64_t)b * BLK; i < e; ++i) x = mix(x, get(i));
      part[b] = x;
    }
    for (uint64_t x : part) h = mix(h, x);
  };
  hash_range([&](uint64_t i) { return g.node_id[i]; }, g.n_nodes());
  hash_range([&](uint64_t i) { return g.label_off[i + 1]; }, g.n_nodes());
  hash_range([&](uint64_t i) { return g.edge_off[i + 1]; }, g.n_nodes());
  hash_range([&](uint64_t i) { return (uint64_t)g.edge_to[i]; }, g.edge_to.size());
  // label bytes, eight at a time
  const uint64_t nw = g.labels.size() / 8;
  hash_range([&](uint64_t i) { uint64_t w; memcpy(&w, g.labels.data() + 8 * i, 8); return w; }, nw);
  for (uint64_t i = nw * 8; i < g.labels.size(); ++i) h = mix(h, (uint8_t)g.labels[i]);
  return h;
}

// Do the index's paths, trims and starting loci lie inside this graph?  (An index file is tied to its
// graph by the fingerprint; this is the structural check behind it: path nodes exist and follow edges,
// head offsets / tail lengths fit their nodes, every starting locus is a base of its node.)
bool index_fits_graph(const Index& x, const Graph& g)
{
  const uint64_t n = g.n_nodes();
  if (x.path_head.size() > x.paths.size() || x.path_tail.size() > x.paths.size()) return false;
  bool ok = true;
#pragma omp parallel for schedule(dynamic, 64) reduction(&& : ok)
  for (int64_t pi = 0; pi < (int64_t)x.paths.size(); ++pi) {
    const auto& P = x.paths[pi];
    bool good = true;
    for (size_t i = 0; good && i < P.size(); ++i) {
      good = P[i] < n;
      if (good && i) {
        const uint32_t u = P[i - 1];
        bool edge = false;
        for (uint64_t e = g.edge_off[u]; e < g.edge_off[u + 1] && !edge; ++e) edge = g.edge_to[e] == P[i];
        good = edge;
      }
    }
    if (good && !P.empty()) {
      const uint32_t hd = (size_t)pi < x.path_head.size() ? x.path_head[pi] : 0;
      const uint32_t tl = (size_t)pi < x.path_tail.size() ? x.path_tail[pi] : 0;
      good = hd <= g.node_len(P.front()) && tl <= g.node_len(P.back());
    }
    ok = ok && good;
  }
  if (!ok || x.loci_node.size() != x.loci_off.size()) return false;
  bool lok = true;
#pragma omp parallel for reduction(&& : lok)
  for (int64_t i = 0; i < (int64_t)x.loci_node.size(); ++i)
    lok = lok && x.loci_node[i] < n && x.loci_off[i] < g.node_len(x.loci_node[i]);
  return lok;
}

Index* build_index(const Graph& g, const psigpu_index_opts& opts,
                   const std::vector<std::vector<uint32_t>>& paths, const std::vector<uint32_t>& head,
                   const std::vector<uint32_t>& tail, int* status, std::string* err)
{
  const uint32_t k = opts.seed_len, step = opts.locus_step;
  uint32_t sa_rate = opts.sa_rate;
  const bool keep = opts.keep_text_sa != 0;
  if (k == 0 || k > PSIGPU_MAX_SEED_LEN) { *status = PSIGPU_ERR_ARG; *err = "seed length out of range"; return nullptr; }
  if (sa_rate == 0) sa_rate = 1;      // 288 GB of HBM: keep the whole suffix array while the text is < 2^31
  if (sa_rate & (sa_rate - 1)) { *status = PSIGPU_ERR_ARG; *err = "sa_rate must be a power of two"; return nullptr; }
  for (auto& P : paths)
    for (size_t i = 0; i + 1 < P.size(); ++i) {
      bool found = false;
      for (uint64_t e = g.edge_off[P[i]]; e < g.edge_off[P[i] + 1]; ++e)
        if (g.edge_to[e] == P[i + 1]) { found = true; break; }
      if (!found) { *status = PSIGPU_ERR_ARG; *err = "path step without an edge"; return nullptr; }
    }
  if ((!head.empty() && head.size() != paths.size()) || (!tail.empty() && tail.size() != paths.size())) {
    *status = PSIGPU_ERR_ARG; *err = "head / tail arrays must have one entry per path"; return nullptr;
  }
  bool trimmed = false;
  for (size_t p = 0; p < paths.size(); ++p) {
    if (paths[p].empty()) continue;
    const uint64_t l0 = g.node_len(paths[p].front()), l1 = g.node_len(paths[p].back());
    const uint32_t h = head.empty() ? 0 : head[p], t = tail.empty() ? 0 : tail[p];
    if (h > l0 || t > l1 || (paths[p].size() == 1 && t && h >= t)) {
      *status = PSIGPU_ERR_ARG; *err = "path head offset / tail length out of range"; return nullptr;
    }
    trimmed = trimmed || h || (t && t < l1);
  }
  Index* x = new Index;
  x->k = k; x->sa_rate = sa_rate; x->context = opts.context;
  x->paths = paths;
  x->path_head = head.empty() ? std::vector<uint32_t>(paths.size(), 0) : head;
  x->path_tail = tail.empty() ? std::vector<uint32_t>(paths.size(), 0) : tail;
  x->locus_step = step ? step : 1;
  x->graph_fp = graph_fingerprint(g);

  // ---- parts: the paths in order, a new part whenever the text would pass the row limit ------------
  // host SA-IS works on int32 indices; the device builder and the index layout on u32
  const uint64_t hard_max = opts.build_on_device ? 0xFFFFFF00ull : 0x7FFFFFF0ull;
  const uint64_t max_text = opts.max_part_text ? std::min<uint64_t>(opts.max_part_text, hard_max) : hard_max;
  std::vector<size_t> cuts{ 0 };
  {
    uint64_t cur = 1;
    for (size_t pi = 0; pi < paths.size(); ++pi) {
      uint64_t len = 1;                                   // the path's text (trimmed ends left out) + its separator
      for (uint32_t v : paths[pi]) len += g.node_len(v);
      if (!paths[pi].empty()) {
        len -= x->path_head[pi];
        if (x->path_tail[pi]) len -= g.node_len(paths[pi].back()) - x->path_tail[pi];
      }
      if (len + 1 >= hard_max) {
        *status = PSIGPU_ERR_ARG;
        *err = opts.build_on_device ? "one indexed path is too long for the 32-bit index layout"
                                    : "indexed text too long for the host suffix sorter (build on the device)";
        delete x; return nullptr;
      }
      if (cur + len >= max_text && cuts.back() != pi) { cuts.push_back(pi); cur = 1; }
      cur += len;
    }
    cuts.push_back(paths.size());
  }
  const size_t n_parts = cuts.size() - 1;
  const bool trace = getenv("PSIGPU_TRACE") != nullptr;
  auto t_mark = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[psigpu] index build: %s %.2f s\n", what, std::chrono::duration<double>(now - t_mark).count());
    t_mark = now;
  };
  if (n_parts > 1 && keep) {
    *status = PSIGPU_ERR_ARG; *err = "an index in several parts cannot keep its text"; delete x; return nullptr;
  }
  if (n_parts > PSIGPU_MAX_PARTS) { *status = PSIGPU_ERR_ARG; *err = "indexed text too long (too many parts)"; delete x; return nullptr; }
  {
    // every part is a complete FM index: searched one after the other in the FM modes, tabulated together for the k-mer table
    int st = build_part(g, opts, sa_rate, keep, paths, x->path_head, x->path_tail, cuts[0], cuts[1], x, err);
    if (st != PSIGPU_OK) { *status = st; delete x; return nullptr; }
    for (size_t pt = 1; pt < n_parts; ++pt) {
      x->more.emplace_back();
      st = build_part(g, opts, sa_rate, false, paths, x->path_head, x->path_tail, cuts[pt], cuts[pt + 1], &x->more.back(), err);
      if (st != PSIGPU_OK) { *status = st; delete x; return nullptr; }
    }
  }

  lap("text, suffix array and tables of the parts");
  // the device routine tracks coverage with one bit per path: full, simple (no node twice) paths, at most 64
  bool bits_suffice = !trimmed && paths.size() <= 64;
  if (bits_suffice && opts.build_on_device) {
    std::vector<uint32_t> seen(g.n_nodes(), 0xFFFFFFFFu);
    for (size_t p = 0; p < paths.size() && bits_suffice; ++p)
      for (uint32_t v : paths[p]) {
        if (seen[v] == (uint32_t)p) { bits_suffice = false; break; }
        seen[v] = (uint32_t)p;
      }
  }
  if (opts.build_on_device && bits_suffice) {
    int st = gpu_find_starting_loci(g, paths, k, step, (int)opts.build_on_device - 1, x->loci_node, x->loci_off, err);
    if (st != PSIGPU_OK) { *status = st; delete x; return nullptr; }
  } else if (opts.build_on_device && !getenv("PSIGPU_HOST_LOCI")) {
    // trimmed paths (patches), many paths, paths that come back to a node: coverage by path STEPS, on the device as on the
    // host (round 5); a node with more steps or longer candidate lists than a thread's pool holds sends the lot to the host
    uint64_t n_hard = 0;
    int st = gpu_find_starting_loci_steps(g, paths, x->path_head, x->path_tail, k, step, (int)opts.build_on_device - 1, x->loci_node,
                                          x->loci_off, &n_hard, err);
    if (st != PSIGPU_OK) { *status = st; delete x; return nullptr; }
    if (trace) fprintf(stderr, "[psigpu]   starting loci by path steps on the device: %llu node(s) left to the host\n", (unsigned long long)n_hard);
    if (n_hard) find_starting_loci(g, paths, x->path_head, x->path_tail, k, step, x->loci_node, x->loci_off);
  } else {
    find_starting_loci(g, paths, x->path_head, x->path_tail, k, step, x->loci_node, x->loci_off);
  }
  lap("starting loci");
  // PSIGPU_BUILD_VERIFY=1 (load campaigns: tools/fuzz_par.sh): what the device made is made again on the host and
  // compared array by array -- a build that differs fails, naming the array and the first place, instead of
  // surfacing later as a hit set that is one record short
  if (opts.build_on_device && getenv("PSIGPU_BUILD_VERIFY")) {
    std::string what;
    auto differ = [&](const char* name, const void* a, size_t na, const void* b, size_t nb, size_t elem) {
      if (!what.empty()) return;
      if (na != nb) { what = std::string(name) + ": " + std::to_string(na) + " against " + std::to_string(nb) + " elements on the host"; return; }
      if (na == 0 || memcmp(a, b, na * elem) == 0) return;
      size_t i = 0;
      while (memcmp((const char*)a + i * elem, (const char*)b + i * elem, elem) == 0) ++i;
      what = std::string(name) + " differs from the host's at element " + std::to_string(i) + " of " + std::to_string(na);
    };
    psigpu_index_opts ho = opts;
    ho.build_on_device = 0;
    for (size_t pt = 0; pt < n_parts && what.empty(); ++pt) {
      Index h;
      const Index& d = pt ? x->more[pt - 1] : *x;
      std::string herr;
      if (build_part(g, ho, sa_rate, false, paths, x->path_head, x->path_tail, cuts[pt], cuts[pt + 1], &h, &herr) != PSIGPU_OK) {
        what = "host build of the part failed: " + herr;      // (e.g. a text the host suffix sorter cannot take)
        break;
      }
      differ("rank blocks", d.blocks.data(), d.blocks.size(), h.blocks.data(), h.blocks.size(), sizeof(RankBlock));
      differ("suffix array samples", d.samples.data(), d.samples.size(), h.samples.data(), h.samples.size(), 4);
      differ("exception rows", d.exc_row.data(), d.exc_row.size(), h.exc_row.data(), h.exc_row.size(), 4);
      differ("exception positions", d.exc_sa.data(), d.exc_sa.size(), h.exc_sa.data(), h.exc_sa.size(), 4);
      differ("exception super-block counts", d.exc_super.data(), d.exc_super.size(), h.exc_super.data(), h.exc_super.size(), 4);
      differ("interval table", d.ftab.data(), d.ftab.size(), h.ftab.data(), h.ftab.size(), 4);
      differ("4-bit text", d.text4.data(), d.text4.size(), h.text4.data(), h.text4.size(), 8);
      differ("C", d.C, 4, h.C, 4, 8);
    }
    if (what.empty()) {
      std::vector<uint32_t> hn, hoff;
      find_starting_loci(g, paths, x->path_head, x->path_tail, k, step, hn, hoff);
      differ("starting loci (node)", x->loci_node.data(), x->loci_node.size(), hn.data(), hn.size(), 4);
      differ("starting loci (offset)", x->loci_off.data(), x->loci_off.size(), hoff.data(), hoff.size(), 4);
    }
    if (!what.empty()) {
      fprintf(stderr, "[psigpu] PSIGPU_BUILD_VERIFY: device index build: %s\n", what.c_str());
      *status = PSIGPU_ERR_DEVICE; *err = "device index build failed verification: " + what; delete x; return nullptr;
    }
  }
  *status = PSIGPU_OK;
  return x;
}

// ------------------------------------------------------------------------------------
// Serialisation: one little-endian container `<prefix>.psigpu`.
// ------------------------------------------------------------------------------------
namespace {
const char MAGIC[8] = { 'P', 'S', 'I', 'G', 'P', 'U', '0', '7' };

template <typename T> bool wr(FILE* f, const std::vector<T>& v)
{
  uint64_t n = v.size();
  if (fwrite(&n, 8, 1, f) != 1) return false;
  return n == 0 || fwrite(v.data(), sizeof(T), n, f) == n;
}
// bytes of the file behind the read position (a length field of an untrusted file is checked against it BEFORE
// anything is allocated)
uint64_t bytes_left(FILE* f)
{
  struct stat st;
  const off_t at = ftello(f);
  if (at < 0 || fstat(fileno(f), &st) != 0 || st.st_size < at) return 0;
  return (uint64_t)(st.st_size - at);
}
template <typename T> bool rd(FILE* f, std::vector<T>& v)
{
  uint64_t n;
  if (fread(&n, 8, 1, f) != 1) return false;
  if (n > bytes_left(f) / sizeof(T)) return false;
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace

int save_index(const Index& x, const std::string& prefix)
{
  FILE* f = fopen((prefix + ".psigpu").c_str(), "wb");
  if (!f) return PSIGPU_ERR_IO;
  bool ok = fwrite(MAGIC, 8, 1, f) == 1;
  uint64_t hdr[8] = { x.k, x.sa_rate, x.context | ((uint64_t)x.ftab_len << 32), x.n, x.C[0], x.C[1], x.C[2], x.C[3] };
  ok = ok && fwrite(hdr, 8, 8, f) == 8;
  uint64_t np = x.paths.size();
  ok = ok && fwrite(&np, 8, 1, f) == 1;
  for (auto& p : x.paths) ok = ok && wr(f, p);
  uint64_t extra[2] = { x.locus_step, x.graph_fp };
  ok = ok && fwrite(extra, 8, 2, f) == 2 && wr(f, x.path_head) && wr(f, x.path_tail);
  ok = ok && wr(f, x.blocks) && wr(f, x.samples) && wr(f, x.exc_row) && wr(f, x.exc_sa) &&
       wr(f, x.seg_start) && wr(f, x.seg_node) && wr(f, x.seg_noff) && wr(f, x.seg_dir) &&
       wr(f, x.loci_node) && wr(f, x.loci_off) && wr(f, x.ftab) && wr(f, x.text4) && wr(f, x.exc_super);
  uint64_t xs0 = x.exc_shift;
  ok = ok && fwrite(&xs0, 8, 1, f) == 1;
  // further parts: text length, C, interval-table length, then the part's arrays
  uint64_t n_more = x.more.size();
  ok = ok && fwrite(&n_more, 8, 1, f) == 1;
  for (const Index& m : x.more) {
    uint64_t ph[6] = { m.n, m.C[0], m.C[1], m.C[2], m.C[3], m.ftab_len };
    ok = ok && fwrite(ph, 8, 6, f) == 6 && wr(f, m.blocks) && wr(f, m.samples) && wr(f, m.exc_row) && wr(f, m.exc_sa) &&
         wr(f, m.seg_start) && wr(f, m.seg_node) && wr(f, m.seg_noff) && wr(f, m.seg_dir) && wr(f, m.ftab) && wr(f, m.text4) &&
         wr(f, m.exc_super);
    uint64_t xs = m.exc_shift;
    ok = ok && fwrite(&xs, 8, 1, f) == 1;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? PSIGPU_OK : PSIGPU_ERR_IO;
}

// every array length of a part follows from its text length
static bool part_consistent(const Index& x, uint32_t sa_rate)
{
  const uint64_t n = x.n;
  bool ok = n >= 1 && n < 0xFFFFFFF0ull && (x.blocks.empty() ? x.exc_row.empty() : x.blocks.size() == n / BLOCK_SYMS + 1) &&
            x.samples.size() == (n + sa_rate - 1) / sa_rate && x.exc_row.size() == x.exc_sa.size() &&
            (x.blocks.empty() || x.exc_super.size() == ((x.blocks.size() - 1) >> x.exc_shift) + 1) &&
            x.seg_start.size() == x.seg_node.size() + 1 && x.seg_noff.size() == x.seg_node.size() &&
            x.seg_dir.size() == (n >> DIR_SHIFT) + 1 && x.ftab_len <= 16 &&
            (x.ftab.empty() ? x.ftab_len == 0 : x.ftab.size() == (2ull << (2 * x.ftab_len))) &&
            (x.text4.empty() || x.text4.size() == n / 16 + 2);
  for (uint32_t d : x.seg_dir) ok = ok && d < x.seg_node.size();
  for (size_t i = 0; ok && i + 1 < x.seg_start.size(); ++i) ok = x.seg_start[i] <= x.seg_start[i + 1];
  return ok && (x.seg_start.empty() || x.seg_start.back() == n);
}

Index* load_index(const std::string& prefix, int* status)
{
  FILE* f = fopen((prefix + ".psigpu").c_str(), "rb");
  if (!f) { *status = PSIGPU_ERR_IO; return nullptr; }
  struct Closer { FILE*& f; ~Closer() { if (f) fclose(f); } } closer{ f };      // (an allocation below may throw)
  std::unique_ptr<Index> x(new Index);
  char magic[8];
  uint64_t hdr[8], np = 0;
  bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, MAGIC, 8) == 0 &&
            fread(hdr, 8, 8, f) == 8 && fread(&np, 8, 1, f) == 1 && np < (1ull << 32) && np <= bytes_left(f) / 8;
  if (ok) {
    x->k = (uint32_t)hdr[0]; x->sa_rate = (uint32_t)hdr[1]; x->context = (uint32_t)hdr[2]; x->ftab_len = (uint32_t)(hdr[2] >> 32);
    x->n = hdr[3];
    for (int i = 0; i < 4; ++i) x->C[i] = hdr[4 + i];
    x->paths.resize(np);
    for (auto& p : x->paths) ok = ok && rd(f, p);
    uint64_t extra[2] = { 0, 0 };
    ok = ok && fread(extra, 8, 2, f) == 2 && rd(f, x->path_head) && rd(f, x->path_tail);
    x->locus_step = (uint32_t)extra[0]; x->graph_fp = extra[1];
    ok = ok && rd(f, x->blocks) && rd(f, x->samples) && rd(f, x->exc_row) && rd(f, x->exc_sa) &&
         rd(f, x->seg_start) && rd(f, x->seg_node) && rd(f, x->seg_noff) && rd(f, x->seg_dir) &&
         rd(f, x->loci_node) && rd(f, x->loci_off) && rd(f, x->ftab) && rd(f, x->text4) && rd(f, x->exc_super);
    uint64_t xs0 = 0;
    ok = ok && fread(&xs0, 8, 1, f) == 1 && xs0 <= EXC_SUPER_SHIFT;
    x->exc_shift = (uint32_t)xs0;
    uint64_t n_more = 0;
    ok = ok && fread(&n_more, 8, 1, f) == 1 && n_more < PSIGPU_MAX_PARTS;
    for (uint64_t i = 0; ok && i < n_more; ++i) {
      x->more.emplace_back();
      Index& m = x->more.back();
      uint64_t ph[6];
      ok = fread(ph, 8, 6, f) == 6;
      if (!ok) break;
      m.n = ph[0]; m.sa_rate = x->sa_rate; m.ftab_len = (uint32_t)ph[5];
      for (int c = 0; c < 4; ++c) m.C[c] = ph[1 + c];
      ok = rd(f, m.blocks) && rd(f, m.samples) && rd(f, m.exc_row) && rd(f, m.exc_sa) && rd(f, m.seg_start) &&
           rd(f, m.seg_node) && rd(f, m.seg_noff) && rd(f, m.seg_dir) && rd(f, m.ftab) && rd(f, m.text4) && rd(f, m.exc_super);
      uint64_t xs = 0;
      ok = ok && fread(&xs, 8, 1, f) == 1 && xs <= EXC_SUPER_SHIFT;
      m.exc_shift = (uint32_t)xs;
    }
  }
  fclose(f); f = nullptr;
  // a corrupt or truncated file must not reach the device: every array length follows from the header
  if (ok) {
    ok = x->k >= 1 && x->k <= PSIGPU_MAX_SEED_LEN && x->sa_rate && !(x->sa_rate & (x->sa_rate - 1)) &&
         part_consistent(*x, x->sa_rate) && x->loci_node.size() == x->loci_off.size() &&
         x->path_head.size() == x->paths.size() && x->path_tail.size() == x->paths.size();
    for (const Index& m : x->more) ok = ok && part_consistent(m, x->sa_rate);
    x->fm_ok = !x->blocks.empty();
    for (Index& m : x->more) m.fm_ok = !m.blocks.empty();
  }
  if (!ok) { *status = PSIGPU_ERR_FORMAT; return nullptr; }
  *status = PSIGPU_OK;
  return x.release();
}

}  // namespace psigpu
