// Starting loci by path steps: the per-node routine of build_gpu.hip's k_steps_loci_* kernels.  Plain C++ that compiles
// for the device (hipcc) and for the host (g++): the CPU suite runs the very function the kernels run over the fixtures
// and compares it with index.cpp's find_starting_loci (psigpu_debug_loci_by_steps, capi_host.cpp) -- there is no GPU
// where the CPU suite runs, and a stack machine with hand-managed lists is the kind of code that wants a cheap check.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define PSI_HD __host__ __device__
#else
#define PSI_HD
#endif

namespace psigpu {

constexpr int STEPS_LOCI_DEPTH = 4 * 63 + 3;       // index.cpp's explore(): depth <= 4 k (k <= PSIGPU_MAX_SEED_LEN)

struct StepGraph {
  const uint64_t* edge_off; const uint32_t* edge_to; const uint32_t* len; const uint32_t* child;
  const uint32_t* step_node; const uint32_t* step_lo; const uint32_t* step_hi; const uint8_t* step_last;
  const uint32_t* at_off; const uint32_t* at;          // steps at every node, ascending
  uint64_t n; uint32_t k, step;
};
constexpr uint32_t SL_POOL = 384, SL_HEADS = 16, SL_AT_NODE = 128;      // per thread: candidate steps in flight, distinct head offsets
                                                                       // at a node, steps at a node -- more: the node is the host's


PSI_HD inline uint32_t sl_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
PSI_HD inline uint32_t sl_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
PSI_HD inline void sl_mark(uint64_t& unc, uint32_t lo, uint32_t hi)       // bits lo .. hi (< 64)
{
  if (hi < lo) return;
  hi = sl_min(hi, 63u);
  if (lo > 63u) return;
  unc |= (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
}

// the loci of node v, in offset order: counted (out_n == nullptr) or written; *hard: this node is the host's
PSI_HD inline uint32_t steps_loci_of_node(const StepGraph& g, uint64_t v, uint32_t* out_n, uint32_t* out_o, bool* hard)
{
  const uint32_t len = g.len[v], k = g.k, child = g.child[v];
  if (len == 0) return 0;
  const uint32_t* sv = g.at + g.at_off[v];
  const uint32_t nsv = g.at_off[v + 1] - g.at_off[v];
  if (nsv > SL_AT_NODE) { *hard = true; return 0; }
  // distinct head offsets of the steps at v, ascending
  uint32_t heads[SL_HEADS];
  uint32_t nh = 0;
  for (uint32_t i = 0; i < nsv; ++i) {
    const uint32_t h = g.step_lo[sv[i]];
    uint32_t at = 0;
    while (at < nh && heads[at] < h) ++at;
    if (at < nh && heads[at] == h) continue;
    if (nh == SL_HEADS) { *hard = true; return 0; }
    for (uint32_t j = nh; j > at; --j) heads[j] = heads[j - 1];
    heads[at] = h; ++nh;
  }
  uint64_t unc_of[SL_HEADS];
  const uint32_t kmax = k - 1;
  struct Frame { uint32_t e, e_end, S, st_off, st_n; };
  Frame st[STEPS_LOCI_DEPTH];
  uint32_t pool[SL_POOL];
  for (uint32_t h = 0; h < nh; ++h) {
    uint64_t unc = 0;
    // walks that leave v: candidates spell v to its end and go on
    uint32_t n0 = 0;
    for (uint32_t i = 0; i < nsv; ++i) {
      const uint32_t s = sv[i];
      if (g.step_lo[s] <= heads[h] && g.step_hi[s] == len && !g.step_last[s]) pool[n0++] = s;      // (nsv <= SL_AT_NODE <= SL_POOL)
    }
    if (n0 == 0) { if (child) sl_mark(unc, 1, sl_min(kmax, child)); unc_of[h] = unc; continue; }
    int sp = 0;
    uint32_t top = n0;                          // pool[0, top) is in use
    st[sp++] = Frame{ (uint32_t)g.edge_off[v], (uint32_t)g.edge_off[v + 1], 0u, 0u, n0 };
    while (sp) {
      Frame& f = st[sp - 1];
      if (f.e == f.e_end) { top = f.st_off; --sp; continue; }      // (the frame's list is the topmost: released with it)
      const uint32_t e = f.e++;
      const uint32_t depth = (uint32_t)sp - 1;
      if (depth > 4 * k) continue;              // guards cycles of empty nodes
      const uint32_t u = g.edge_to[e];
      const uint32_t ulen = g.len[u];
      // the candidate steps that go on through u
      uint32_t n_next = 0, hi_max = 0;
      bool over = false;
      for (uint32_t i = 0; i < f.st_n; ++i) {
        const uint32_t t = pool[f.st_off + i] + 1;       // (a candidate is never the last step of its path)
        if (g.step_node[t] != u) continue;
        hi_max = sl_max(hi_max, g.step_hi[t]);
        if (!g.step_last[t]) { if (top + n_next >= SL_POOL) { over = true; break; } pool[top + n_next++] = t; }
      }
      if (over) { *hard = true; return 0; }
      // needs that end inside u: covered while some candidate spells that many of u's bases
      if (ulen && f.S + hi_max < kmax && hi_max < ulen) sl_mark(unc, f.S + hi_max + 1, sl_min(kmax, f.S + ulen));
      const uint32_t S2 = f.S + ulen;
      if (S2 >= kmax) continue;
      if (n_next == 0) { const uint32_t cu = g.child[u]; if (cu) sl_mark(unc, S2 + 1, sl_min(kmax, S2 + cu)); continue; }
      if (sp >= STEPS_LOCI_DEPTH) continue;           // (cannot happen: depth <= 4 k + 1 < STEPS_LOCI_DEPTH)
      st[sp++] = Frame{ (uint32_t)g.edge_off[u], (uint32_t)g.edge_off[u + 1], S2, top, n_next };
      top += n_next;
    }
    unc_of[h] = unc;
  }
  uint32_t since = 0, cnt = 0;
  for (uint32_t o = 0; o < len; ++o) {
    if ((uint64_t)len - o + child < k) continue;            // no k-walk starts here
    bool take;
    const int64_t need = (int64_t)k - (int64_t)(len - o);
    if (need <= 0) {
      // the k-walk lies inside v: covered iff one step spells [o, o + k)
      take = true;
      for (uint32_t i = 0; i < nsv && take; ++i)
        if (g.step_lo[sv[i]] <= o && (uint64_t)g.step_hi[sv[i]] >= (uint64_t)o + k) take = false;
    } else {
      // the range of o: the last head offset <= o (none: no candidate at all)
      uint32_t h = 0;
      while (h < nh && heads[h] <= o) ++h;
      take = h == 0 ? true : ((unc_of[h - 1] >> need) & 1) != 0;
    }
    if (!take) continue;
    if (since % g.step == 0) {
      if (out_n) { out_n[cnt] = (uint32_t)v; out_o[cnt] = o; }
      ++cnt;
    }
    ++since;
  }
  return cnt;
}


}  // namespace psigpu
