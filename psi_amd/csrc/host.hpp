// Host-side objects behind the C ABI (include/psi_gpu.h): graph store, path selection,
// FM-index construction, starting-loci detection, (de)serialisation.  No HIP here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <algorithm>
#include <atomic>
#include <thread>

#include <sys/mman.h>

#include "../../include/psi_gpu.h"

namespace psigpu {

// Text symbol coding used by the builder and the suffix sorter.
enum : uint8_t { SYM_END = 0, SYM_SEP = 1, SYM_A = 2, SYM_C = 3, SYM_G = 4, SYM_T = 5 };

constexpr uint32_t BLOCK_SYMS = 192;       // symbols per 64-byte rank block
constexpr uint32_t DIR_SHIFT = 6;          // one directory entry per 64 text positions
constexpr uint32_t NO_NODE = 0xFFFFFFFFu;

constexpr uint32_t EXC_SUPER_SHIFT = 16;   // rank blocks per exception super-block: 2^16 (12.6 M rows)

struct RankBlock {            // 64 bytes, one HBM sector
  uint32_t cnt[3];            // A, C, G before this block (exceptions not counted)
  uint32_t exc;               // (exceptions before this block, counted from the start of its super-block of
                              // 2^exc_shift blocks) << 8 | min(255, exceptions inside); the exceptions in front of
                              // every super-block are a side array (Index::exc_super), so that a text may hold any
                              // number of separators -- the 44.6 M patches of a whole genome x 3 walks -- in 24 bits
  uint64_t sym[6];            // 192 symbols as bit planes: for group g of 64 symbols, word 2g holds
                              // their low bits and word 2g+1 their high bits (symbol j at bit j % 64)
};
static_assert(sizeof(RankBlock) == 64, "rank block must be one 64-byte sector");

struct Graph {
  std::vector<uint64_t> node_id;      // external ids, rank order = file order
  std::vector<uint64_t> label_off;    // n+1
  std::string labels;
  std::vector<uint64_t> edge_off;     // n+1
  std::vector<uint32_t> edge_to;
  std::vector<std::vector<uint32_t>> paths;   // embedded paths as node ranks
  std::vector<std::string> path_names;

  uint64_t n_nodes() const { return node_id.size(); }
  uint64_t node_len(uint32_t v) const { return label_off[v + 1] - label_off[v]; }
};

struct Index {
  uint32_t k = 0, sa_rate = 0, context = 0;
  std::vector<std::vector<uint32_t>> paths;    // indexed paths (node ranks)
  std::vector<uint32_t> path_head, path_tail;  // per path: offset of its first indexed base in its first node; indexed
                                               // bases of its last node (0 = all) -- Path::left / right of a patch
  bool fm_ok = true;                           // rank blocks / exceptions / interval table present
  uint32_t locus_step = 1;                     // psikt -e the starting loci were sampled with
  uint64_t graph_fp = 0;                       // fingerprint of the graph the index was made for
  uint64_t n = 0;                              // text length
  std::vector<RankBlock> blocks;
  uint64_t C[4] = { 0, 0, 0, 0 };
  std::vector<uint32_t> samples, exc_row, exc_sa;
  std::vector<uint32_t> exc_super;             // exceptions in front of every super-block of 2^exc_shift rank blocks
  uint32_t exc_shift = EXC_SUPER_SHIFT;
  uint32_t ftab_len = 0;
  std::vector<uint32_t> ftab;                  // 2 x 4^ftab_len
  std::vector<uint64_t> text4;                 // 4 bits / symbol, first symbol in the top nibble
  std::vector<uint32_t> seg_start, seg_node, seg_noff, seg_dir;
  std::vector<uint32_t> loci_node, loci_off;
  std::vector<uint8_t> text;                   // kept only on request
  std::vector<int32_t> sa;                     // kept only on request
  // An index whose text would pass the 32-bit row limit is made in several PARTS: consecutive groups
  // of paths, each a complete FM index of its own -- text, rank blocks, interval table, suffix array,
  // segment table (the fields above; paths, loci and the scalars live in the first part only).  A seed is
  // searched in every part (a pattern never spans two paths, so the parts' occurrences are disjoint and
  // their union is the answer).  Parts beyond the first:
  std::vector<Index> more;
};

// graph.cpp
Graph* load_graph_file(const std::string& path, int* status, std::string* err, uint32_t flags = 0);

// pathsel.cpp / index.cpp
// pathsel.cpp: Haplotyper walks and patches
void pick_paths(const Graph& g, uint32_t n_per_region, bool patched, uint32_t context, uint64_t rng_seed,
                std::vector<std::vector<uint32_t>>& out, std::vector<uint32_t>& head, std::vector<uint32_t>& tail);
// index.cpp
uint64_t graph_fingerprint(const Graph& g);
bool index_fits_graph(const Index& x, const Graph& g);
Index* build_index(const Graph& g, const psigpu_index_opts& opts,
                   const std::vector<std::vector<uint32_t>>& paths, const std::vector<uint32_t>& head,
                   const std::vector<uint32_t>& tail, int* status, std::string* err);
// Fault the pages of a freshly reserved range in from all threads (madvise on page-aligned slices; where the
// kernel does not know MADV_POPULATE_WRITE the call fails and the first writer faults them in, as before).
// v.resize( n ) of a whole-genome array is otherwise a page fault per 4 KB on ONE thread: seconds per array.
inline void populate_pages(void* p, size_t bytes)
{
#ifdef MADV_POPULATE_WRITE
  const uintptr_t PAGE = 4096, SLICE = 64u << 20;
  const uintptr_t a = ((uintptr_t)p + PAGE - 1) & ~(PAGE - 1), b = ((uintptr_t)p + bytes) & ~(PAGE - 1);
  if (b <= a || b - a < (8u << 20)) return;
  const uint64_t n = (b - a + SLICE - 1) / SLICE;
  unsigned hw = std::thread::hardware_concurrency();
  const unsigned nt = (unsigned)std::min<uint64_t>(n, std::min<unsigned>(hw ? hw : 1, 32));
  std::atomic<uint64_t> next{ 0 };
  auto work = [&] {
    for (uint64_t i; (i = next.fetch_add(1)) < n;) {
      const uintptr_t lo = a + (uintptr_t)i * SLICE, hi = lo + SLICE < b ? lo + SLICE : b;
      (void)madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
#else
  (void)p; (void)bytes;
#endif
}

// v.resize( n ) (value-initialised) with the pages faulted in by all threads first
template <typename V>
inline void resize_populated(V& v, size_t n)
{
  if (n > v.capacity()) { v.reserve(n); populate_pages(v.data(), n * sizeof(typename V::value_type)); }
  v.resize(n);
}

void find_starting_loci(const Graph& g, const std::vector<std::vector<uint32_t>>& paths,
                        const std::vector<uint32_t>& path_head, const std::vector<uint32_t>& path_tail,
                        uint32_t k, uint32_t step, std::vector<uint32_t>& loci_node,
                        std::vector<uint32_t>& loci_off);
int save_index(const Index& x, const std::string& prefix);

// refio.cpp: the reference's `<prefix>_paths` file -> paths (node ranks) and their trims
int read_reference_paths(const std::string& file, const Graph& g, uint64_t* context, bool* forward,
                         std::vector<std::vector<uint32_t>>& paths, std::vector<uint32_t>& head,
                         std::vector<uint32_t>& tail, std::string* err);

// build_gpu.hip: suffix array + FM arrays on the device (same results as the host path)
int gpu_build_fm(const std::vector<uint8_t>& T, uint32_t sa_rate, uint32_t q, int device, Index* x,
                 std::vector<int32_t>* sa_out, std::string* err);

// build_gpu.hip: starting loci on the device, same result as find_starting_loci()
int gpu_find_starting_loci(const Graph& g, const std::vector<std::vector<uint32_t>>& paths, uint32_t k, uint32_t step,
                           int device, std::vector<uint32_t>& loci_node, std::vector<uint32_t>& loci_off, std::string* err);
// ... for any set of paths (trimmed / patches, more than 64, nodes visited twice); *n_hard != 0: left to the host
int gpu_find_starting_loci_steps(const Graph& g, const std::vector<std::vector<uint32_t>>& paths,
                                 const std::vector<uint32_t>& path_head, const std::vector<uint32_t>& path_tail,
                                 uint32_t k, uint32_t step, int device, std::vector<uint32_t>& loci_node,
                                 std::vector<uint32_t>& loci_off, uint64_t* n_hard, std::string* err);
int gpu_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, uint64_t n,
                       unsigned end_bit, std::string* err);
int gpu_running_max_u64(uint64_t* data, uint64_t n, std::string* err);

// hits.cpp: parallel sort-unique of hit records by (read_id, read_offset, node_id, node_offset) on the
// host: only for records that do not fit the device sorter's 64-bit key (HitSorter::fits)
uint64_t sort_unique_hits(psigpu_hit* hits, uint64_t n);

// hits_gpu.hip: the same on the device -- records packed into 64-bit keys, radix sort, unique, expand
struct HitSorter {
  void* keys_a = nullptr; void* keys_b = nullptr; void* tmp = nullptr;
  uint64_t cap = 0; size_t tmp_cap = 0;
  HitSorter() = default;
  HitSorter(const HitSorter&) = delete;
  HitSorter& operator=(const HitSorter&) = delete;
  ~HitSorter();
  // can a hit of this chunk / graph be packed into one 64-bit key?
  static bool fits(uint64_t n_reads, uint64_t max_read_len, uint64_t n_nodes, uint64_t max_node_len);
  // d_in[0..n) -> d_out[0..*d_count) sorted and unique; asynchronous on `stream` (a hipStream_t);
  // d_count is a device pointer.  Read ids lie in [rec_offset, rec_offset + n_reads).
  int run(const psigpu_hit* d_in, uint64_t n, uint64_t rec_offset, uint64_t n_reads, uint64_t max_read_len,
          uint64_t n_nodes, uint64_t max_node_len, bool id_affine, uint64_t id_base, const uint64_t* d_ids_sorted,
          psigpu_hit* d_out, uint64_t* d_count, void* stream, std::string* err);
  // Hits that come out seed by seed are in (read, read offset) order already: only the hits of one seed
  // -- one or two, almost always -- can be out of order or equal.  Sorts every such group of up to 32 in
  // place and leaves 0 in *d_flag when the array is then sorted and duplicate-free (the answer of run(),
  // without the radix sort), non-zero when it is not (a longer group, a duplicate, hits not grouped by
  // seed): the caller falls back to run().  Asynchronous on `stream`.  With `d_n` the number of hits is
  // min(*d_n, n) (a count that is still on the device) and *d_flag is expected to be zero already.
  static int fix_grouped(psigpu_hit* d_hits, uint64_t n, const unsigned long long* d_n, uint64_t* d_flag, void* stream,
                         std::string* err);
};
Index* load_index(const std::string& prefix, int* status);

}  // namespace psigpu

struct psigpu_graph { psigpu::Graph g; };
struct psigpu_index {
  psigpu::Index x;
  mutable std::vector<psigpu_index_view> more_views;      // views of x.more, handed out by psigpu_index_view_get
};
