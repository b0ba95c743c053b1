// C entry points of the host side (graph loading, index construction); see
// include/psi_gpu.h for what each one replaces in the reference.
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "host.hpp"
#include "loci_steps.hpp"
#include "sais.hpp"

using namespace psigpu;

static thread_local std::string g_host_err;

extern "C" {

uint32_t psigpu_abi_version(void) { return PSIGPU_ABI_VERSION; }

const char* psigpu_host_last_error(void) { return g_host_err.c_str(); }

psigpu_graph* psigpu_graph_load(const char* path, int* status) { return psigpu_graph_load_opts(path, 0, status); }

psigpu_graph* psigpu_graph_load_opts(const char* path, uint32_t flags, int* status)
{
  int st = PSIGPU_OK;
  std::string err;
  Graph* g = path ? load_graph_file(path, &st, &err, flags) : nullptr;
  if (!path) st = PSIGPU_ERR_ARG;
  if (status) *status = st;
  if (!g) { g_host_err = err; return nullptr; }
  psigpu_graph* h = new psigpu_graph;
  h->g = std::move(*g);
  delete g;
  return h;
}

psigpu_graph* psigpu_graph_from_csr(uint64_t n_nodes, const uint64_t* node_id,
                                    const uint64_t* label_off, const char* labels,
                                    const uint64_t* edge_off, const uint32_t* edge_to,
                                    uint64_t n_paths, const uint64_t* path_off,
                                    const uint32_t* path_nodes, int* status)
{
  auto fail = [&](const char* msg) -> psigpu_graph* {
    g_host_err = msg;
    if (status) *status = PSIGPU_ERR_ARG;
    return nullptr;
  };
  if (n_nodes >= 0xFFFFFFF0ull) return fail("too many nodes");
  if (n_nodes && (!node_id || !label_off || !edge_off)) return fail("null graph arrays");
  psigpu_graph* h = new psigpu_graph;
  Graph& g = h->g;
  g.node_id.assign(node_id, node_id + n_nodes);
  if (n_nodes) {
    g.label_off.assign(label_off, label_off + n_nodes + 1);
    g.edge_off.assign(edge_off, edge_off + n_nodes + 1);
  } else {
    g.label_off.assign(1, 0);
    g.edge_off.assign(1, 0);
  }
  for (uint64_t i = 0; i < n_nodes; ++i)
    if (g.label_off[i + 1] < g.label_off[i] || g.edge_off[i + 1] < g.edge_off[i]) {
      delete h;
      return fail("offsets must be non-decreasing");
    }
  if (g.label_off[n_nodes]) g.labels.assign(labels, labels + g.label_off[n_nodes]);
  for (auto& c : g.labels) c = (char)toupper((unsigned char)c);
  if (g.edge_off[n_nodes]) g.edge_to.assign(edge_to, edge_to + g.edge_off[n_nodes]);
  for (uint32_t t : g.edge_to)
    if (t >= n_nodes) { delete h; return fail("edge target out of range"); }
  for (uint64_t p = 0; p < n_paths; ++p) {
    std::vector<uint32_t> nodes(path_nodes + path_off[p], path_nodes + path_off[p + 1]);
    for (uint32_t v : nodes)
      if (v >= n_nodes) { delete h; return fail("path node out of range"); }
    g.paths.push_back(std::move(nodes));
    g.path_names.push_back("path" + std::to_string(p));
  }
  if (status) *status = PSIGPU_OK;
  return h;
}

void psigpu_graph_free(psigpu_graph* g) { delete g; }

int psigpu_graph_view_get(const psigpu_graph* h, psigpu_graph_view* out)
{
  if (!h || !out) return PSIGPU_ERR_ARG;
  const Graph& g = h->g;
  out->n_nodes = g.n_nodes();
  out->node_id = g.node_id.data();
  out->label_off = g.label_off.data();
  out->labels = g.labels.data();
  out->edge_off = g.edge_off.data();
  out->edge_to = g.edge_to.data();
  return PSIGPU_OK;
}

uint64_t psigpu_graph_path_count(const psigpu_graph* h) { return h ? h->g.paths.size() : 0; }
uint64_t psigpu_graph_edge_count(const psigpu_graph* h) { return h ? h->g.edge_to.size() : 0; }

uint64_t psigpu_graph_path(const psigpu_graph* h, uint64_t i, uint32_t* out, uint64_t cap)
{
  if (!h || i >= h->g.paths.size()) return 0;
  const auto& p = h->g.paths[i];
  uint64_t n = p.size() < cap ? p.size() : cap;
  if (out && n) memcpy(out, p.data(), n * 4);
  return p.size();
}

static psigpu_index* wrap_index(Index* x, int st, const std::string& err, int* status)
{
  if (status) *status = st;
  if (!x) { g_host_err = err; return nullptr; }
  psigpu_index* h = new psigpu_index;
  h->x = std::move(*x);
  delete x;
  return h;
}

psigpu_index* psigpu_index_build(const psigpu_graph* g, const psigpu_index_opts* opts, int* status)
{
  if (!g || !opts) { if (status) *status = PSIGPU_ERR_ARG; return nullptr; }
  if (opts->n_per_region && g->g.paths.empty()) {
    // SeedFinder::pick_paths: "no reference path found in the input graph"
    // (include/psi/seed_finder.hpp:1145-1147)
    g_host_err = "no reference path found in the input graph";
    if (status) *status = PSIGPU_ERR_ARG;
    return nullptr;
  }
  // SeedFinder::set_context (seed_finder.hpp:1772-1787): no context without patching; patching with
  // context 0 means context = seed length
  psigpu_index_opts o = *opts;
  if (!o.patched) o.context = 0;
  else if (o.context == 0) o.context = o.seed_len;
  std::vector<std::vector<uint32_t>> paths;
  std::vector<uint32_t> head, tail;
  const auto t_pick = std::chrono::steady_clock::now();
  pick_paths(g->g, o.n_per_region, o.patched != 0, o.context, o.rng_seed, paths, head, tail);
  if (getenv("PSIGPU_TRACE"))
    fprintf(stderr, "[psigpu] index build: path picking %.2f s (%zu paths / patches)\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pick).count(), paths.size());
  int st; std::string err;
  Index* x = build_index(g->g, o, paths, head, tail, &st, &err);
  return wrap_index(x, st, err, status);
}

psigpu_index* psigpu_index_build_paths(const psigpu_graph* g, const psigpu_index_opts* opts,
                                       uint64_t n_paths, const uint64_t* path_off,
                                       const uint32_t* path_nodes, int* status)
{
  return psigpu_index_build_patches(g, opts, n_paths, path_off, path_nodes, nullptr, nullptr, status);
}

psigpu_index* psigpu_index_build_patches(const psigpu_graph* g, const psigpu_index_opts* opts,
                                         uint64_t n_paths, const uint64_t* path_off,
                                         const uint32_t* path_nodes, const uint32_t* head_off,
                                         const uint32_t* tail_len, int* status)
{
  if (!g || !opts || (n_paths && (!path_off || !path_nodes))) { if (status) *status = PSIGPU_ERR_ARG; return nullptr; }
  std::vector<std::vector<uint32_t>> paths;
  // (the caller guarantees that path_nodes holds path_off[n_paths] entries: the API cannot bound it)
  for (uint64_t p = 0; p < n_paths; ++p)
    if (path_off[p + 1] < path_off[p]) {
      g_host_err = "path offsets must not decrease";
      if (status) *status = PSIGPU_ERR_ARG;
      return nullptr;
    }
  for (uint64_t p = 0; p < n_paths; ++p) {
    std::vector<uint32_t> nodes(path_nodes + path_off[p], path_nodes + path_off[p + 1]);
    for (uint32_t v : nodes)
      if (v >= g->g.n_nodes()) {
        g_host_err = "path node out of range";
        if (status) *status = PSIGPU_ERR_ARG;
        return nullptr;
      }
    paths.push_back(std::move(nodes));
  }
  std::vector<uint32_t> head, tail;
  if (head_off) head.assign(head_off, head_off + n_paths);
  if (tail_len) tail.assign(tail_len, tail_len + n_paths);
  int st; std::string err;
  Index* x = build_index(g->g, *opts, paths, head, tail, &st, &err);
  return wrap_index(x, st, err, status);
}

// An index over the paths a reference-written `<prefix>_paths` file holds (refio.cpp): the paths and their trims are
// read, the FM index and the starting loci are made here (the file's own node-id index and the companion
// `<prefix>` file -- an sdsl::csa_wt of the reversed text -- are not needed for that).
psigpu_index* psigpu_index_from_reference_paths(const psigpu_graph* g, const psigpu_index_opts* opts, const char* paths_file,
                                                uint64_t* context_out, uint32_t* forward_out, int* status)
{
  if (!g || !opts || !paths_file) { if (status) *status = PSIGPU_ERR_ARG; return nullptr; }
  std::vector<std::vector<uint32_t>> paths;
  std::vector<uint32_t> head, tail;
  uint64_t context = 0;
  bool forward = false;
  std::string err;
  // nothing may leave an extern "C" function by exception: a file this reader cannot make sense of, or one whose
  // (checked) sizes still exhaust memory, is an error code
  try {
    int st = read_reference_paths(paths_file, g->g, &context, &forward, paths, head, tail, &err);
    if (st != PSIGPU_OK) { g_host_err = err; if (status) *status = st; return nullptr; }
    if (context_out) *context_out = context;
    if (forward_out) *forward_out = forward ? 1u : 0u;
    psigpu_index_opts o = *opts;
    o.context = (uint32_t)std::min<uint64_t>(context, 0xFFFFFFFFull);        // PathIndex::load_paths_set takes the file's (pathindex.hpp:286-289)
    Index* x = build_index(g->g, o, paths, head, tail, &st, &err);
    return wrap_index(x, st, err, status);
  } catch (const std::bad_alloc&) {
    g_host_err = std::string(paths_file) + ": out of memory";
    if (status) *status = PSIGPU_ERR_NOMEM;
  } catch (const std::exception& e) {
    g_host_err = std::string(paths_file) + ": " + e.what();
    if (status) *status = PSIGPU_ERR_FORMAT;
  }
  return nullptr;
}

void psigpu_index_free(psigpu_index* x) { delete x; }

static void fill_view(const Index& x, const Index& first, psigpu_index_view* v)
{
  v->seed_len = first.k; v->sa_rate = x.sa_rate; v->context = first.context;
  v->n_paths = (uint32_t)first.paths.size();
  v->text_len = x.n;
  v->n_blocks = x.blocks.size();
  v->bwt_blocks = x.blocks.empty() ? nullptr : x.blocks.data();      // none: an index that can only be tabulated
  for (int i = 0; i < 4; ++i) v->C[i] = x.C[i];
  v->n_samples = x.samples.size(); v->sa_samples = x.samples.data();
  v->n_exc = x.exc_row.size(); v->exc_row = x.exc_row.data(); v->exc_sa = x.exc_sa.data();
  v->ftab_len = x.ftab_len; v->exc_shift = x.exc_shift; v->ftab = x.ftab.empty() ? nullptr : x.ftab.data();
  v->exc_super = x.exc_super.empty() ? nullptr : x.exc_super.data();
  v->text4 = x.text4.empty() ? nullptr : x.text4.data();
  v->n_segs = x.seg_node.size();
  v->seg_start = x.seg_start.data(); v->seg_node = x.seg_node.data(); v->seg_noff = x.seg_noff.data();
  v->n_dir = x.seg_dir.size(); v->seg_dir = x.seg_dir.data();
  v->n_loci = x.loci_node.size(); v->loci_node = x.loci_node.data(); v->loci_off = x.loci_off.data();
  v->n_more_parts = 0; v->reserved2 = 0; v->more_parts = nullptr;
}

int psigpu_index_view_get(const psigpu_index* h, psigpu_index_view* v)
{
  if (!h || !v) return PSIGPU_ERR_ARG;
  fill_view(h->x, h->x, v);
  h->more_views.resize(h->x.more.size());
  for (size_t i = 0; i < h->x.more.size(); ++i) fill_view(h->x.more[i], h->x, &h->more_views[i]);
  v->n_more_parts = (uint32_t)h->more_views.size();
  v->more_parts = h->more_views.empty() ? nullptr : h->more_views.data();
  return PSIGPU_OK;
}

int psigpu_index_save(const psigpu_index* x, const char* prefix)
{
  if (!x || !prefix) return PSIGPU_ERR_ARG;
  return save_index(x->x, prefix);
}

psigpu_index* psigpu_index_load(const char* prefix, int* status)
{
  int st = PSIGPU_ERR_ARG;
  Index* x = nullptr;
  try {
    x = prefix ? load_index(prefix, &st) : nullptr;
  } catch (const std::exception&) {               // (sizes are checked against the file; memory can still run out)
    st = PSIGPU_ERR_NOMEM;
  }
  return wrap_index(x, st, "cannot load index", status);
}

uint64_t psigpu_index_path_count(const psigpu_index* x) { return x ? x->x.paths.size() : 0; }

uint64_t psigpu_index_path(const psigpu_index* h, uint64_t i, uint32_t* out, uint64_t cap)
{
  if (!h || i >= h->x.paths.size()) return 0;
  const auto& p = h->x.paths[i];
  uint64_t n = p.size() < cap ? p.size() : cap;
  if (out && n) memcpy(out, p.data(), n * 4);
  return p.size();
}

int psigpu_index_path_trim(const psigpu_index* h, uint64_t i, uint32_t* head_off, uint32_t* tail_len)
{
  if (!h || i >= h->x.paths.size()) return PSIGPU_ERR_ARG;
  if (head_off) *head_off = i < h->x.path_head.size() ? h->x.path_head[i] : 0;
  if (tail_len) *tail_len = i < h->x.path_tail.size() ? h->x.path_tail[i] : 0;
  return PSIGPU_OK;
}

int psigpu_index_matches(const psigpu_index* h, const psigpu_graph* g, uint32_t seed_len, uint32_t locus_step)
{
  if (!h || !g) return 0;
  if (locus_step == 0) locus_step = 1;
  return h->x.k == seed_len && h->x.locus_step == locus_step && h->x.graph_fp == graph_fingerprint(g->g) &&
         index_fits_graph(h->x, g->g);
}

uint32_t psigpu_index_locus_step(const psigpu_index* h) { return h ? h->x.locus_step : 0; }

// Test hook (not part of include/psi_gpu.h): the starting loci of the index's paths by the routine the DEVICE runs for
// trimmed / many / non-simple paths (loci_steps.hpp, build_gpu.hip k_steps_loci_*), run here on the host over structures
// made the way the kernels make them.  Returns the number of loci (written to out_node / out_off while they fit `cap`),
// -1 when a node is beyond the routine's per-thread pool (the device build then takes the host routine), -2 on bad arguments.
int64_t psigpu_debug_loci_by_steps(const psigpu_graph* gh, const psigpu_index* h, uint32_t locus_step, uint32_t* out_node,
                                   uint32_t* out_off, uint64_t cap)
{
  if (!gh || !h) return -2;
  const Graph& g = gh->g;
  const Index& x = h->x;
  if (locus_step == 0) locus_step = 1;
  const uint64_t n = g.n_nodes();
  const uint32_t k = x.k;
  std::vector<uint32_t> len(n), reach(n), child(n, 0);
  for (uint64_t v = 0; v < n; ++v) { len[v] = (uint32_t)g.node_len((uint32_t)v); reach[v] = std::min<uint32_t>(k, len[v]); }
  for (bool changed = true; changed;) {
    changed = false;
    for (uint64_t v = n; v-- > 0;) {
      uint32_t best = 0;
      for (uint64_t e = g.edge_off[v]; e < g.edge_off[v + 1]; ++e) best = std::max(best, reach[g.edge_to[e]]);
      child[v] = best;
      const uint32_t r = (uint32_t)std::min<uint64_t>(k, (uint64_t)len[v] + best);
      if (r > reach[v]) { reach[v] = r; changed = true; }
    }
  }
  std::vector<uint64_t> first(x.paths.size() + 1, 0);
  for (size_t p = 0; p < x.paths.size(); ++p) first[p + 1] = first[p] + x.paths[p].size();
  const uint64_t total = first.back();
  std::vector<uint32_t> step_node(total + 1), lo(total + 1), hi(total + 1), at_off(n + 1, 0), at(total + 1);
  std::vector<uint8_t> last(total + 1, 0);
  for (size_t p = 0; p < x.paths.size(); ++p)
    for (size_t i = 0; i < x.paths[p].size(); ++i) {
      const uint64_t s = first[p] + i;
      step_node[s] = x.paths[p][i]; lo[s] = 0; hi[s] = len[step_node[s]];
    }
  for (size_t p = 0; p < x.paths.size(); ++p) {
    if (x.paths[p].empty()) continue;
    const uint64_t s0 = first[p], s1 = first[p + 1] - 1;
    if (x.path_head.size() >= x.paths.size()) lo[s0] = std::min(len[step_node[s0]], x.path_head[p]);
    if (x.path_tail.size() >= x.paths.size() && x.path_tail[p]) hi[s1] = std::min(len[step_node[s1]], x.path_tail[p]);
    last[s1] = 1;
  }
  for (uint64_t s = 0; s < total; ++s) ++at_off[step_node[s] + 1];
  for (uint64_t v = 0; v < n; ++v) at_off[v + 1] += at_off[v];
  {
    std::vector<uint32_t> fill(at_off.begin(), at_off.end() - 1);
    for (uint64_t s = 0; s < total; ++s) at[fill[step_node[s]]++] = (uint32_t)s;      // (in step order: ascending per node)
  }
  std::vector<uint64_t> edge_off(g.edge_off.begin(), g.edge_off.end());
  StepGraph sg = { edge_off.data(), g.edge_to.data(), len.data(), child.data(), step_node.data(), lo.data(), hi.data(), last.data(),
                   at_off.data(), at.data(), n, k, locus_step };
  uint64_t count = 0;
  std::vector<uint32_t> bn(1 << 16), bo(1 << 16);
  for (uint64_t v = 0; v < n; ++v) {
    bool hard = false;
    const uint32_t c = steps_loci_of_node(sg, v, nullptr, nullptr, &hard);
    if (hard) return -1;
    if (c == 0) continue;
    if (c > bn.size()) { bn.resize(c); bo.resize(c); }
    steps_loci_of_node(sg, v, bn.data(), bo.data(), &hard);
    for (uint32_t i = 0; i < c; ++i, ++count)
      if (count < cap && out_node && out_off) { out_node[count] = bn[i]; out_off[count] = bo[i]; }
  }
  return (int64_t)count;
}

// The starting loci for another locus step, recomputed from the index's own paths and trims: nothing that
// lies beside the index file is trusted for this (a `<prefix>_loci_e<E>l<K>` file carries no graph
// fingerprint and no trace of the paths it was made for).
int psigpu_index_set_locus_step(psigpu_index* h, const psigpu_graph* g, uint32_t locus_step)
{
  if (!h || !g) return PSIGPU_ERR_ARG;
  if (locus_step == 0) locus_step = 1;
  Index& x = h->x;
  if (x.graph_fp != graph_fingerprint(g->g) || !index_fits_graph(x, g->g)) {
    g_host_err = "the index was not made for this graph";
    return PSIGPU_ERR_ARG;
  }
  if (x.locus_step == locus_step) return PSIGPU_OK;
  find_starting_loci(g->g, x.paths, x.path_head, x.path_tail, x.k, locus_step, x.loci_node, x.loci_off);
  x.locus_step = locus_step;
  return PSIGPU_OK;
}

// `<prefix>_loci_e<E>l<K>`: the reference's starting-loci file (SeedFinder::save_starts / open_starts,
// seed_finder.hpp:1640-1679; get_sloci_filepath; psi::serialize of a container, utils.hpp:521-588):
// u64 count, then `count` raw psi::Position<> = { gum id_type node id, gum offset_type offset } with
// EXTERNAL node ids (coordinate_id on the way out, id_by_coordinate on the way in).  gum is not in the
// reference tree; its GraphBaseTrait types are taken as int64_t / uint64_t (16 bytes per locus), and a
// file whose size does not fit that layout is rejected.
static std::string loci_path(const char* prefix, uint32_t k, uint32_t step)
{
  return std::string(prefix) + "_loci_e" + std::to_string(step) + "l" + std::to_string(k);
}

int psigpu_loci_save(const psigpu_index* h, const psigpu_graph* g, const char* prefix)
{
  if (!h || !g || !prefix) return PSIGPU_ERR_ARG;
  const Index& x = h->x;
  FILE* f = fopen(loci_path(prefix, x.k, x.locus_step).c_str(), "wb");
  if (!f) return PSIGPU_ERR_IO;
  uint64_t n = x.loci_node.size();
  bool ok = fwrite(&n, 8, 1, f) == 1;
  for (uint64_t i = 0; ok && i < n; ++i) {
    if (x.loci_node[i] >= g->g.n_nodes()) { ok = false; break; }
    int64_t id = (int64_t)g->g.node_id[x.loci_node[i]];
    uint64_t off = x.loci_off[i];
    ok = fwrite(&id, 8, 1, f) == 1 && fwrite(&off, 8, 1, f) == 1;
  }
  ok = (fclose(f) == 0) && ok;
  return ok ? PSIGPU_OK : PSIGPU_ERR_IO;
}

int psigpu_loci_load(psigpu_index* h, const psigpu_graph* g, const char* prefix, uint32_t locus_step)
{
  if (!h || !g || !prefix) return PSIGPU_ERR_ARG;
  if (locus_step == 0) locus_step = 1;
  Index& x = h->x;
  FILE* f = fopen(loci_path(prefix, x.k, locus_step).c_str(), "rb");
  if (!f) return PSIGPU_ERR_IO;
  uint64_t n = 0;
  bool ok = fread(&n, 8, 1, f) == 1;
  if (ok) {
    fseek(f, 0, SEEK_END);
    ok = (uint64_t)ftell(f) == 8 + 16 * n;           // the assumed Position<> layout
    fseek(f, 8, SEEK_SET);
  }
  std::vector<std::pair<uint32_t, uint32_t>> loci;
  if (ok) {
    std::unordered_map<uint64_t, uint32_t> rank;
    rank.reserve(g->g.n_nodes() * 2);
    for (uint64_t v = 0; v < g->g.n_nodes(); ++v) rank.emplace(g->g.node_id[v], (uint32_t)v);
    loci.reserve(n);
    for (uint64_t i = 0; ok && i < n; ++i) {
      int64_t id; uint64_t off;
      ok = fread(&id, 8, 1, f) == 1 && fread(&off, 8, 1, f) == 1;
      if (!ok) break;
      auto it = rank.find((uint64_t)id);
      ok = it != rank.end() && off < g->g.node_len(it->second);
      if (ok) loci.emplace_back(it->second, (uint32_t)off);
    }
  }
  fclose(f);
  if (!ok) { g_host_err = "not a starting-loci file of this graph"; return PSIGPU_ERR_FORMAT; }
  std::sort(loci.begin(), loci.end());            // by node rank, then offset (seed_finder.hpp:1695-1722 groups by node)
  loci.erase(std::unique(loci.begin(), loci.end()), loci.end());
  x.loci_node.resize(loci.size()); x.loci_off.resize(loci.size());
  for (size_t i = 0; i < loci.size(); ++i) { x.loci_node[i] = loci[i].first; x.loci_off[i] = loci[i].second; }
  x.locus_step = locus_step;
  return PSIGPU_OK;
}

const uint8_t* psigpu_index_text(const psigpu_index* x)
{
  return x && !x->x.text.empty() ? x->x.text.data() : nullptr;
}

const int32_t* psigpu_index_sa(const psigpu_index* x)
{
  return x && !x->x.sa.empty() ? x->x.sa.data() : nullptr;
}

#if defined(__x86_64__)
// `n_words` whole words (32 bases each) starting at base index `at` (a multiple of 32), ASCII at `src`: the two code bits
// of a letter are bits 1..2 of its byte (A 00, C 01, G 11, T 10 -> c = y ^ (y >> 1)); the bytes are reversed first so
// that movemask hands out the first base in the top bit, and the two bit planes are interleaved by pdep.
__attribute__((target("avx2,bmi2")))
static uint64_t pack_words_avx2(const char* src, uint64_t n_words, uint64_t* packed, uint64_t* n_mask, uint64_t at, uint64_t* bad)
{
  const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  const __m256i fold = _mm256_set1_epi8((char)0xDF), m3 = _mm256_set1_epi8(3), m1 = _mm256_set1_epi8(1);
  const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  for (uint64_t w = 0; w < n_words; ++w) {
    __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32 * w));
    x = _mm256_shuffle_epi8(x, rev);
    x = _mm256_permute2x128_si256(x, x, 1);                       // byte j now holds base 31 - j
    const __m256i u = _mm256_and_si256(x, fold);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
    const __m256i y = _mm256_and_si256(_mm256_srli_epi16(x, 1), m3);
    __m256i c = _mm256_xor_si256(y, _mm256_and_si256(_mm256_srli_epi16(y, 1), m1));
    c = _mm256_and_si256(c, ok);                                   // (a base that is not ACGT: code 0 and its mask bit)
    const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 7));
    const uint32_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(c, 6));
    packed[w] = _pdep_u64(hi, 0xAAAAAAAAAAAAAAAAull) | _pdep_u64(lo, 0x5555555555555555ull);
    const uint32_t notok = ~(uint32_t)_mm256_movemask_epi8(ok);    // bit j = base 31 - j
    if (notok) {
      uint32_t nat = 0;
      for (uint32_t j = 0; j < 32; ++j) if ((notok >> j) & 1u) nat |= 1u << (31 - j);
      *bad += (uint64_t)__builtin_popcount(nat);
      const uint64_t i = at + 32 * w;
      if (n_mask) n_mask[i >> 6] |= (uint64_t)nat << (i & 63);
    }
  }
  return n_words;
}
#endif

// ASCII bases -> 2-bit words + "not ACGT" bits (layout: psigpu_find_seeds_packed).  Eight bases at a time with the SWAR
// arithmetic the device's ASCII packer uses (k_seed_pack): the two code bits of a letter are bits 1..2 of its byte.
uint64_t psigpu_pack_reads(const char* bases, uint64_t first, uint64_t n, uint64_t* packed, uint64_t* n_mask)
{
  if (!bases || !packed || n == 0) return 0;
  uint64_t bad = 0;
  uint64_t i = first;
  const uint64_t end = first + n;
  auto one = [&](uint64_t at) {
    const unsigned char ch = (unsigned char)bases[at - first];
    const unsigned u = ch & 0xDFu, d = u - 0x41u;
    const bool ok = d < 20u && ((0x80045u >> d) & 1u);
    unsigned c = (u >> 1) & 3u;
    c ^= c >> 1;
    if (ok) packed[at >> 5] |= (uint64_t)c << (62 - 2 * (at & 31));
    else { ++bad; if (n_mask) n_mask[at >> 6] |= 1ull << (at & 63); }
  };
  while (i < end && (i & 31)) one(i++);                     // to a word boundary
#if defined(__x86_64__)
  // 32 bases per step where the host has AVX2 + BMI2 (round 5: the SWAR loop below packs ~1.5 GB/s per thread, 12 ms per
  // 1 M-read chunk on eight threads -- five device calls' worth; this one runs at what the memory delivers)
  static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
  if (have_avx2 && i + 32 <= end) {
    const uint64_t done = pack_words_avx2(bases + (i - first), (end - i) / 32, packed + (i >> 5), n_mask, i, &bad);
    i += 32 * done;
  }
#endif
  for (; i + 32 <= end; i += 32) {
    uint64_t word = 0;
    uint32_t nbits = 0;                                      // mask bits of these 32 bases, base j in bit j
    for (int q = 0; q < 4; ++q) {
      uint64_t x;
      memcpy(&x, bases + (i - first) + 8 * q, 8);            // first base in the low byte
      const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;
      const uint64_t y = (x >> 1) & 0x0303030303030303ull;
      const uint64_t b0 = y & 0x0101010101010101ull, b1 = (y >> 1) & 0x0101010101010101ull, t = b0 & b1;
      const uint64_t e = 0x4141414141414141ull + (b0 << 1) + (b1 << 4) + (b1 << 1) + b1 - (t << 4) + t;
      uint64_t c = y ^ b1;                                   // per byte: A 0, C 1, G 2, T 3
      const uint64_t diff = u ^ e;                           // a non-zero byte = not ACGT
      if (diff) {
        for (int j = 0; j < 8; ++j)
          if ((diff >> (8 * j)) & 0xFF) { nbits |= 1u << (8 * q + j); c &= ~(0xFFull << (8 * j)); ++bad; }
      }
      // gather the eight 2-bit codes, FIRST base (low byte) most significant
      c = __builtin_bswap64(c);                              // first base in the top byte
      c = (c | (c >> 6)) & 0x000F000F000F000Full;
      c = (c | (c >> 12)) & 0x000000FF000000FFull;
      c = (c | (c >> 24)) & 0xFFFFull;
      word |= c << (48 - 16 * q);
    }
    packed[i >> 5] = word;                                   // (a whole word: nobody else writes it)
    if (n_mask && nbits) n_mask[i >> 6] |= (uint64_t)nbits << (i & 63);
  }
  while (i < end) one(i++);
  return bad;
}

int psigpu_suffix_array(const uint8_t* text, uint64_t n, uint32_t sigma, int32_t* sa_out)
{
  if (!text || !sa_out || n == 0 || n >= 0x7FFFFFF0ull || sigma == 0 || sigma > 256) return PSIGPU_ERR_ARG;
  if (text[n - 1] != 0) return PSIGPU_ERR_ARG;
  for (uint64_t i = 0; i + 1 < n; ++i)
    if (text[i] == 0 || text[i] >= sigma) return PSIGPU_ERR_ARG;
  suffix_array(text, sa_out, (int32_t)n, (int32_t)sigma);
  return PSIGPU_OK;
}

}  // extern "C"
