// MI355X (gfx950) seed-finding kernels and the device half of the C ABI.
//
// One call of psigpu_find_seeds* = one chunk of psikt's loop (reference src/psikt.cpp:195-204):
//
//   K0  k_seed_scan_* / k_seed_pack         seeding()                 include/psi/sequence.hpp:1688-1745
//       k_table_insert                      index_reads()             include/psi/seed_finder.hpp:1089-1097
//   K1  k_fm_search[_direct]                kmer_exact_matches descent include/psi/index_iter.hpp:835-841
//                                           -> Iter::go_down           include/psi/fmindex.hpp:851-869
//       k_kmer_probe                        the same result from the k-mer table (PSIGPU_MODE_KMER_TABLE)
//   K2  k_fm_locate_direct / k_fm_walk      get_occurrences + mapping  include/psi/fmindex.hpp:734-777,
//                                                                      include/psi/pathindex.hpp:378-416
//   K4  k_traverse<false>                   TraverserBFS::run          include/psi/traverser_bfs.hpp:72-161
//       k_traverse<true>                    the same walks enumerated once per index (the tables)
//   K5  emission: scan-placed records in K2, private chunks in K4 (callbacks at index_iter.hpp:676,
//       traverser_bfs.hpp:109)
//
// Integer rank / popcount / compare work, HBM-latency and -bandwidth bound; no MFMA.
// Wavefronts are 64 lanes.  A "quad" is 4 adjacent lanes that fetch one 64-byte rank block
// as 4 x 16 B (one coalesced sector) and combine partial popcounts with DPP quad permutes.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <execinfo.h>
#include <memory>
#include <signal.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "host.hpp"

using namespace psigpu;

namespace {

constexpr uint32_t VERIFY_ROWS = 8;           // SA intervals up to this size are finished against the text
constexpr uint32_t PFX_SHORT = 12;            // first-level seed-prefix bitmap: 4^12 bits = 2 MiB (L2-resident)
constexpr uint32_t PFX_LONG = 14;             // second level: 4^14 bits = 32 MiB
constexpr uint64_t KEY_INVALID = ~0ull;      // a valid key uses at most 62 bits
constexpr uint32_t NIL = 0xFFFFFFFFu;

struct NodeRec {          // 32 bytes: one fetch serves a whole hop
  // Short nodes (<= 32 bp) carry an EXTENDED head: their own label followed by the labels of the
  // successor chain for as long as the out-degree is exactly 1, up to 32 bases in all.  `len` is
  // the number of inline bases, and the out-edges stored here are those of the last node the
  // chain covered completely -- or, when the 32-base cap cut a node, that node itself entered at
  // offset `coff`.  Long nodes carry their first 32 bases and are read from the label words.
  uint64_t w0;            // bits 0..39 label offset (bases), 40..55 out-degree, 56..61 coff,
                          // 62 long node, 63 has-N (long nodes)
  uint32_t len;           // inline bases (short) / label length (long)
  uint32_t edge_off;      // out-degree == 2: the SECOND out-edge's target; > 2: offset into edge_to
  uint64_t head2;         // inline bases, 2 bit each, first base most significant
  uint32_t headn;         // N mask of those bases, first base most significant
  uint32_t edge0;         // target of the first out-edge (NIL for sinks)
};
static_assert(sizeof(NodeRec) == 32, "node record must be 32 bytes");

// 16-byte form of the same record for the common case (short node, no N in the inline bases,
// out-degree <= 2, second out-edge within +-32767 ranks of the first): what the traverser
// stages in LDS.  Nodes that do not fit carry the SLOW flag and are read as NodeRec.
struct NodeLite {
  uint64_t head2;         // inline bases (extended head)
  uint32_t edge0;
  uint32_t meta;          // 0..5 inline bases, 6..7 out-degree, 8..13 coff, 14 SLOW, 16..31 edge1 - edge0
};
constexpr uint32_t LITE_SLOW = 1u << 14;

struct TableSlot {        // 16 bytes: one seed-table slot; a probe's sector holds everything
  unsigned long long key; // KEY_INVALID = empty
  uint32_t val;           // first seed with this k-mer
  uint32_t dup;           // chain of further seeds with it (NIL = none)
};

struct SegRec {           // 16 bytes: text segment -> graph position
  uint32_t start;         // text position of the segment's first base
  uint32_t noff;          // node offset of that base
  uint64_t node_id;       // external node id
};

// One record per suffix-array row, made when the index is loaded (whole SA + text resident, seed
// length k and interval-table length q fixed): everything K1 and K2 need to know about a row sits
// in one 16-byte read instead of SA value -> text window -> segment directory -> segment.
struct SaRec {
  uint32_t node;          // rank of the node holding text position SA[row] - (k - q): where a seed whose
  uint32_t noff;          //   last q bases start at SA[row] begins, and the offset in that node
  uint64_t ctx;           // bits 0..57: the 29 text bases in front of SA[row] (the nearest in bits 0..1),
                          // bits 58..62: how many of them are bases of the same path (0..29)
};

struct SeedIv { uint32_t lo, cnt; };

// K1 -> K2, one entry per seed (structure of arrays: written and read coalesced)
struct SeedOut {
  uint32_t* iv_lo;        // first SA row of the seed's interval
  uint32_t* iv_cnt;       // on-path occurrences
  uint32_t* iv_aux;       // verified intervals: bits 0..7 rows that matched, 8..15 bases in front of the rows,
                          // bit 31: (on_node, on_noff) hold the hit of the first matching row
  uint32_t* on_node;
  uint32_t* on_noff;
  uint32_t* off_first;    // locus k-mer table: first entry of the run -- or the node rank when
  uint32_t* off_cnt;      //   OFF_INLINE is set in the count (a single locus, kept in the slot itself)
  uint32_t* off_noff;     //   and its offset
};
constexpr uint32_t OFF_INLINE = 0x80000000u;
constexpr uint32_t AUX_RESOLVED = 0x80000000u;
constexpr uint32_t AUX_ONPOS = 0x40000000u;       // the occurrences are the run on_pos[lo, lo + con) (k-mer table)
constexpr uint64_t LKT_INLINE = 1ull << 63;   // table slot: val = node rank, dup = offset of the k-mer's only locus     // SA interval of a seed, cnt == 0: no occurrence

// Device-side counters, one per 128-byte line: atomics on different counters must not
// serialise behind each other in the same L2 channel.
struct alignas(128) PaddedCounter { unsigned long long v; char pad[120]; };
// Statistics that every wave adds to are striped over 32 lines: atomics on one address retire at
// about one per 11 ns on this part, so 16 K waves adding to a single counter hold a kernel for
// 0.18 ms -- longer than k_seed_pack's real work.  The host adds the stripes.
constexpr int STRIPES = 32;
struct StripedCounter {
  PaddedCounter s[STRIPES];
  __device__ __forceinline__ void add(unsigned long long x) { atomicAdd(&s[blockIdx.x & (STRIPES - 1)].v, x); }
  unsigned long long total() const { unsigned long long t = 0; for (int i = 0; i < STRIPES; ++i) t += s[i].v; return t; }
};
struct DevCounters {
  StripedCounter n_seeds_valid;
  StripedCounter n_live;         // seeds with a non-empty interval
  PaddedCounter n_hits_on;       // on-path hits: total of the per-seed interval sizes
  PaddedCounter n_hits_tab;      // on-path hits + hits from the locus k-mer table (what K2 writes)
  StripedCounter n_kpaths;
  PaddedCounter n_spill;         // append cursor of the spill queue
  PaddedCounter n_chunks;        // traverser output chunks handed out
  PaddedCounter n_hits_off;      // records in those chunks (scan total)
  StripedCounter n_lf_steps;     // LF steps K1 executed (per seed)
  StripedCounter n_rows_verified; // SA rows K1 checked against the text
  StripedCounter n_locate_steps; // LF steps K2 walked to sampled rows (sa_rate > 1)
  PaddedCounter n_defer;         // seeds k_fm_search_direct left to the quad kernel
  PaddedCounter n_seeds_true;    // the scan's seed count (comes back to the host with the counters)
  StripedCounter max_read_len;   // longest read of the chunk, a running maximum per stripe (the hit sorter sizes its key fields with it)
  PaddedCounter not_grouped;     // sort-unique asked for: set when ordering each seed's hits in place was not enough
  PaddedCounter not_uniform;     // PSIGPU_UNIFORM_READS was claimed and a read of the chunk has another length
  PaddedCounter serial;          // the call's serial number, stored by the kernel that zeroes the counters: what comes back to the
                                 // host must carry the serial of THIS call (a stale hand-back is detected, not believed)
  PaddedCounter dbg0, dbg1;      // diagnostics (builds with -DTRAV_STATS)
  PaddedCounter ticket;          // k_kmer_step: the next tile to hand out (workgroups take tiles in the order they start)
  StripedCounter n_hits_on_s;    // ... its on-path hits, added per workgroup (the host adds them to n_hits_on)
  __host__ unsigned long long hits_on() const { return n_hits_on.v + n_hits_on_s.total(); }
};

// Seeds of up to 31 bases are one 64-bit word (2 bits per base, first base most significant) and that is what every
// table and every kernel of the default path is made for.  Seeds of 32..63 bases (psikt takes any -l:
// src/psikt.cpp:327) are 128-bit words through the SAME kernels instantiated for the wider type -- the FM search, the
// traverser and its seed table; the tabulating modes (k-mer table, locus table) stay with one word.
typedef unsigned __int128 u128;
template <typename KEY> __device__ __host__ __forceinline__ constexpr KEY key_invalid() { return ~(KEY)0; }   // (a valid key uses < all bits)

template <typename KEY>
struct TravItemT {        // 16 bytes (32 with 128-bit k-mers)
  KEY kmer;               // marker bit at 2*depth, bases below it (first base most significant)
  uint32_t node;
  uint32_t locus;
};
typedef TravItemT<uint64_t> TravItem;

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// A 16-byte record is ONE load.  Left alone, the compiler fetches the field a branch tests first and the rest of the record
// behind the branch (k_kmer_probe: global_load_dwordx3 + a dependent global_load_dword per look; k_traverse: the node
// record's meta word by a flat load, its bases and edge by a second): two memory latencies in a row per record where
// one request brings the sector.  The empty asm makes all four words live at the point of the load.
__device__ __forceinline__ void keep_whole(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ uint4 load16(const void* p)
{
  uint4 v = *reinterpret_cast<const uint4*>(p);
  keep_whole(v);
  return v;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
  return (1ull << lane_id()) - 1ull;
}

// DPP quad permutes: xor-1, xor-2 butterflies and broadcast of quad lane 0.
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // [1,0,3,2]
}
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // [2,3,0,1]
}
__device__ __forceinline__ uint32_t quad_bcast0(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true);   // [0,0,0,0]
}
__device__ __forceinline__ uint32_t quad_sum(uint32_t v)
{
  v += quad_xor1(v);
  v += quad_xor2(v);
  return v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// position of the highest set bit
__device__ __forceinline__ uint32_t hibit(uint64_t x) { return 63u - (uint32_t)__clzll((long long)x); }
__device__ __forceinline__ uint32_t hibit(u128 x)
{
  const uint64_t h = (uint64_t)(x >> 64);
  return h ? 127u - (uint32_t)__clzll((long long)h) : 63u - (uint32_t)__clzll((long long)(uint64_t)x);
}
// what the chunk's seed table is keyed by: the k-mer itself, or -- two words -- a 64-bit fingerprint of it (the
// lookup then compares the k-mer of every seed it finds: exact whatever the fingerprints do)
__device__ __forceinline__ uint64_t table_key(uint64_t k) { return k; }
__device__ __forceinline__ uint64_t table_key(u128 k)
{
  const uint64_t f = mix64((uint64_t)k ^ mix64((uint64_t)(k >> 64) + 0x9E3779B97F4A7C15ull));
  return f == KEY_INVALID ? 0ull : f;
}

struct FMView {
  const uint4* blocks;       // n_blocks x 4 x 16 B
  const uint32_t* exc_row;   // [n_exc] exception rows, then -- same array -- the exceptions in front of every super-block of
                             // 2^exc_shift rank blocks (a few hundred words): one pointer, and two narrow fields share a
                             // register (the search kernel sits at the SGPR count that still allows 8 waves per SIMD)
  uint32_t n_exc;
  uint32_t n;                // text length
  uint32_t C[4];
  const uint2* ftab;         // [4^ftab_len] SA interval of the q-mer, or nullptr
  uint16_t ftab_len, exc_shift;
  __device__ __forceinline__ uint32_t n_super() const { return ((n / BLOCK_SYMS) >> exc_shift) + 1; }
  const uint64_t* text4;     // the text, 4 bits per symbol (nullptr: never verify against the text)
  const uint32_t* sa;        // whole suffix array when sa_rate == 1, else nullptr
  const SaRec* sarec;        // per-row records for this seed length, or nullptr
};

// exceptions listed for this block that sit below row `i` (rare slow path, quad lane 0 only; the index's
// arrays are passed one by one: a reference to the view would force the whole struct into scratch memory)
__device__ __noinline__ uint32_t exc_below(const uint32_t* __restrict__ rows /* the block's first exception */, uint32_t n, uint32_t i)
{
  uint32_t c = 0;                       // rows are sorted; i lies inside the block, so a row >= i ends the scan
  while (c < n && rows[c] < i) ++c;
  return c;
}

// Do the `rem` (1..16) text symbols in front of position `pos` spell the first `rem` bases of the
// seed (2-bit key of k bases, first base most significant) with no separator among them?
template <typename KEY>
__device__ __forceinline__ bool text_matches(const uint64_t* __restrict__ text4, uint32_t pos, uint32_t rem,
                                             KEY key, uint32_t k)
{
  if (pos < rem) return false;
  uint32_t a = pos - rem, w = a >> 4, sh = (a & 15) * 4;
  uint64_t x = text4[w] << sh;
  if (sh) x |= text4[w + 1] >> (64 - sh);               // 16 nibbles starting at a, first on top
  uint64_t top = rem == 16 ? ~0ull : ~(~0ull >> (4 * rem));
  if (x & top & 0x4444444444444444ull) return false;    // a separator / the sentinel
  uint64_t y = x & 0x3333333333333333ull;               // nibbles -> 2-bit codes, order kept
  y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  y = (y | (y >> 4)) & 0x00FF00FF00FF00FFull;
  y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
  y = (y | (y >> 16)) & 0x00000000FFFFFFFFull;
  uint32_t got = (uint32_t)y >> (32 - 2 * rem);
  uint32_t want = (uint32_t)(key >> (2 * (k - rem)));
  return got == want;
}

// rank_c(i) = #{ j < i : BWT[j] == c }, computed by a quad, branch-free.  `v` is this lane's
// 16-byte chunk of block i/192: lane 0 holds the header, lanes 1..3 hold 64 symbols each as two
// bit planes (v.x|v.y = low bits, v.z|v.w = high bits).  Every lane evaluates the header
// arithmetic on its own chunk (garbage on lanes 1..3) and the quad takes lane 0's result with a
// DPP broadcast; the symbol popcounts of lanes 1..3 are summed with two DPP butterflies.
// The exceptions in front of every super-block of rank blocks are a few hundred words that every rank of a T and
// every exception lookup needs: the LF kernels keep them in LDS (a global load here would sit behind the block's
// and add its latency to the step; measured: the LF search went from 0.83 to 1.05 ms per chr22-like step with it).
constexpr uint32_t SUP_LDS = 352;       // (2^32 / 192) >> 16 = 341 super-blocks at most in the default layout
__device__ __forceinline__ void stage_exc_super(const FMView& fm, uint32_t* s_sup)
{
  const uint32_t n = min(fm.n_super(), SUP_LDS);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_sup[i] = fm.exc_row[fm.n_exc + i];
  __syncthreads();
}
__device__ __forceinline__ uint32_t exc_super(const FMView& fm, const uint32_t* s_sup, uint32_t blk)
{
  const uint32_t i = blk >> fm.exc_shift;
  return i < SUP_LDS ? s_sup[i] : fm.exc_row[fm.n_exc + i];        // (beyond: only the tests' tiny super-blocks)
}

__device__ __forceinline__ uint32_t quad_rank(const FMView& fm, const uint32_t* s_sup, uint4 v, uint32_t ql, uint32_t c,
                                              uint32_t i)
{
  uint32_t blk = i / BLOCK_SYMS, off = i - blk * BLOCK_SYMS;
  // header (meaningful on lane 0 only)
  // T = rows - A - C - G - exceptions in front (header field + what lies in front of the block's super-block:
  // a load that depends on the row alone, not on the block, from an array that stays in cache)
  uint32_t base = 0;
  if (c == 3) base = blk * BLOCK_SYMS - v.x - v.y - v.z - (v.w >> 8) - exc_super(fm, s_sup, blk);
  base = c == 2 ? v.z : base;
  base = c == 1 ? v.y : base;
  base = c == 0 ? v.x : base;
  if (ql == 0 && c == 0 && (v.w & 0xFF) != 0)
  {
    const uint32_t e = (v.w >> 8) + exc_super(fm, s_sup, blk), ne = v.w & 0xFF;
    base -= exc_below(fm.exc_row + e, ne == 255 ? fm.n_exc - e : ne, i);
  }
  base = quad_bcast0(base);
  // symbols: this lane covers [64 (ql-1), 64 ql); m = how many of them lie below `off`
  int32_t rel = (int32_t)off - (int32_t)(ql * 64) + 64;
  uint32_t m = ql == 0 ? 0u : (uint32_t)min(max(rel, 0), 64);
  uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
  uint64_t hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
  uint64_t eq = (lo ^ ((c & 1u) ? 0ull : ~0ull)) & (hi ^ ((c & 2u) ? 0ull : ~0ull));
  uint64_t mask = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
  uint32_t part = (uint32_t)__popcll(eq & mask);
  return base + quad_sum(part);
}

// Counters / counts go back to the host through a kernel that stores into mapped pinned memory, not
// through a copy-engine transfer: a 20-KB D2H queues behind whatever large copy the same SDMA
// engine is busy with (the hits of the previous sub-batch in the host entry's pipeline), and the
// compute stream would then wait for it.
__global__ void __launch_bounds__(256) k_publish(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16)
{
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// read offsets of a sub-batch, straight from the caller's (pinned) array: out[i] = in[i] - in[0]
__global__ void __launch_bounds__(256) k_rebase_offsets(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n)
{
  const uint64_t b0 = in[0];
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i] - b0;
}

// ------------------------------------------------------------------------------------
// K0: seeding
// ------------------------------------------------------------------------------------
// exclusive scan of u32 counts into u64 offsets: 3 phases, 4096 items per block
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_tiles(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_sum)
{
  __shared__ uint64_t sh[SCAN_THREADS];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = SCAN_THREADS / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_sums(uint64_t* tile_sum, uint64_t n_tiles, uint64_t* total)
{
  // one workgroup walks the tile sums 256 at a time with a running carry
  __shared__ uint64_t sh[SCAN_THREADS];
  uint64_t carry = 0;
  for (uint64_t base = 0; base < n_tiles; base += SCAN_THREADS) {
    uint64_t i = base + threadIdx.x;
    uint64_t v = i < n_tiles ? tile_sum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < SCAN_THREADS; d <<= 1) {
      uint64_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < n_tiles) tile_sum[i] = carry + sh[threadIdx.x] - v;
    carry += sh[SCAN_THREADS - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_final(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ tile_sum,
             uint64_t* __restrict__ out)
{
  __shared__ uint64_t sh[SCAN_THREADS];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += in[base + i];
  sh[threadIdx.x] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 256 per-thread sums
  for (int d = 1; d < SCAN_THREADS; d <<= 1) {
    uint64_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint64_t run = tile_sum[blockIdx.x] + sh[threadIdx.x] - s;
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) { out[base + i] = run; run += in[base + i]; }
  if (base <= n && n < base + SCAN_ITEMS) out[n] = run;     // out has n+1 entries
}

// The same three-phase scan over the reads' seed counts, computed from the read offsets on the
// fly (no count array, no separate count kernel); the last phase also leaves the proportional
// guess ratio of k_seed_pack.
__device__ __forceinline__ uint32_t seeds_of_read(const uint64_t* __restrict__ read_off, uint64_t r, uint32_t k, uint32_t step)
{
  // offsets 0, step, 2 step ... while i < len - k + 1 (sequence.hpp:1711-1714); reads shorter
  // than k give none
  uint64_t len = read_off[r + 1] - read_off[r];
  return len >= k ? (uint32_t)((len - k) / step + 1) : 0u;
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_seed_scan_tiles(const uint64_t* __restrict__ read_off, uint64_t n, uint32_t k, uint32_t step, uint64_t* __restrict__ tile_sum,
                  DevCounters* __restrict__ ctr, unsigned long long serial)
{
  __shared__ uint64_t sh[SCAN_THREADS];
  if (blockIdx.x == 0) {                       // first kernel of a call: it also zeroes the call's counters
    uint4* z = reinterpret_cast<uint4*>(ctr);
    for (uint32_t i = threadIdx.x; i < sizeof(DevCounters) / 16; i += SCAN_THREADS)
      z[i] = (i == offsetof(DevCounters, serial) / 16) ? make_uint4((uint32_t)serial, (uint32_t)(serial >> 32), 0, 0) : make_uint4(0, 0, 0, 0);
  }
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < n) s += seeds_of_read(read_off, base + i, k, step);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = SCAN_THREADS / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(SCAN_THREADS)
k_seed_scan_final(const uint64_t* __restrict__ read_off, uint64_t n, uint32_t k, uint32_t step,
                  const uint64_t* __restrict__ tile_sum /* raw, from k_seed_scan_tiles */, uint64_t* __restrict__ out,
                  uint64_t* __restrict__ params, DevCounters* __restrict__ ctr)
{
  __shared__ uint64_t sh[SCAN_THREADS];
  __shared__ uint64_t shp[SCAN_THREADS / 64];
  // seeds in the tiles before this one: every workgroup adds them up for itself (a few hundred
  // values) instead of waiting for a one-workgroup kernel in between
  uint64_t before = 0;
  for (uint64_t i = threadIdx.x; i < blockIdx.x; i += SCAN_THREADS) before += tile_sum[i];
  for (int d = 32; d > 0; d >>= 1) before += __shfl_down(before, d);
  if ((threadIdx.x & 63) == 0) shp[threadIdx.x >> 6] = before;
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t c[SCAN_ITEMS];
  uint64_t s = 0, longest = 0;
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    c[i] = 0;
    if (base + i < n) {
      const uint64_t len = read_off[base + i + 1] - read_off[base + i];
      c[i] = len >= k ? (uint32_t)((len - k) / step + 1) : 0u;
      longest = max(longest, len);
    }
    s += c[i];
  }
  // (one atomic per wave on ONE address would cost more than the scan itself: 11 ns each)
  for (int d = 32; d > 0; d >>= 1) longest = max(longest, (uint64_t)__shfl_down(longest, d));
  if ((threadIdx.x & 63) == 0 && longest)
    atomicMax(&ctr->max_read_len.s[(blockIdx.x * 4 + (threadIdx.x >> 6)) & (STRIPES - 1)].v, (unsigned long long)longest);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < SCAN_THREADS; d <<= 1) {
    uint64_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  before = 0;
  for (int w = 0; w < SCAN_THREADS / 64; ++w) before += shp[w];
  uint64_t run = before + sh[threadIdx.x] - s;
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) { out[base + i] = run; run += c[i]; }
  if (base <= n && n < base + SCAN_ITEMS) out[n] = run;     // out has n+1 entries
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) {
    // params[0] = number of seeds, params[1] = 2^64 * n_reads / n_seeds (the proportional guess of k_seed_pack)
    const uint64_t total = before + sh[SCAN_THREADS - 1];
    unsigned __int128 r = total ? ((unsigned __int128)n << 64) / total : 0;
    params[0] = total;
    params[1] = r > (unsigned __int128)~0ull ? ~0ull : (uint64_t)r;
    ctr->n_seeds_true.v = total;
  }
}

static const bool env_plain_stores = getenv("PSIGPU_PLAIN_STORES") != nullptr;      // A/B: k_kmer_emit without the transposed record stores

// PSIGPU_UNIFORM_READS: every read has the same length, so a seed's read and offset follow from its number -- no scan of
// the reads' seed counts, no per-seed search for the owning read.  The first kernel of such a call: the counters zeroed,
// the call's serial number, the seed count and the longest read where the scan kernels would have left them.
__global__ void __launch_bounds__(256)
k_seed_init_uniform(DevCounters* __restrict__ ctr, unsigned long long serial, uint64_t* __restrict__ params, uint64_t n_seeds,
                    uint64_t read_len)
{
  uint4* z = reinterpret_cast<uint4*>(ctr);
  for (uint32_t i = threadIdx.x; i < sizeof(DevCounters) / 16; i += 256) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i == offsetof(DevCounters, serial) / 16) v = make_uint4((uint32_t)serial, (uint32_t)(serial >> 32), 0, 0);
    if (i == offsetof(DevCounters, n_seeds_true) / 16) v = make_uint4((uint32_t)n_seeds, (uint32_t)(n_seeds >> 32), 0, 0);
    if (i == offsetof(DevCounters, max_read_len) / 16) v = make_uint4((uint32_t)read_len, (uint32_t)(read_len >> 32), 0, 0);
    z[i] = v;
  }
  if (threadIdx.x == 0) { params[0] = n_seeds; params[1] = 0; }
}

// ASCII base -> 2-bit code (A 0, C 1, G 2, T 3, either case), -1 for anything else; branch-free
__device__ __forceinline__ int base2(char ch)
{
  uint32_t u = (uint32_t)(unsigned char)ch & 0xDFu;        // fold case
  uint32_t d = u - 0x41u;                                   // 'A' -> 0, 'C' -> 2, 'G' -> 6, 'T' -> 19
  bool ok = d < 20u && ((0x80045u >> d) & 1u);
  uint32_t c = (u >> 1) & 3u;                               // A 00, C 01, G 11, T 10
  c ^= c >> 1;                                              // A 0, C 1, G 2, T 3
  return ok ? (int)c : -1;
}

// 8 bases (one unaligned 64-bit load, first base in the low byte) -> 2-bit codes of the first `take`
// of them, first base most significant; ok is cleared when one of them is not ACGT (either case)
__device__ __forceinline__ uint64_t pack8(uint64_t x, uint32_t take, uint32_t& ok)
{
  x = __builtin_bswap64(x);                       // first base in the top byte
  if (take < 8) x = (x >> (8 * (8 - take))) | (0x4141414141414141ull << (8 * take));
  const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;   // fold case
  const uint64_t y = (x >> 1) & 0x0303030303030303ull;      // bits 1..2 of a letter: A 00, C 01, G 11, T 10
  const uint64_t b0 = y & 0x0101010101010101ull, b1 = (y >> 1) & 0x0101010101010101ull, t = b0 & b1;
  // the letter those two bits stand for, per byte: 0x41 + 2 b0 + 0x13 b1 - 0x0F (b0 & b1) = A, C, T (b1), G (both);
  // all eight bytes are ACGT iff they equal it (no carry crosses a byte: every byte stays in 0x41..0x54)
  const uint64_t e = 0x4141414141414141ull + (b0 << 1) + (b1 << 4) + (b1 << 1) + b1 - (t << 4) + t;
  ok &= (u == e);
  uint64_t c = y ^ b1;                            // A 0, C 1, G 2, T 3
  c = (c | (c >> 6)) & 0x000F000F000F000Full;
  c = (c | (c >> 12)) & 0x000000FF000000FFull;
  c = (c | (c >> 24)) & 0xFFFFull;
  return c;
}

// one thread per seed: 2-bit key (first base most significant); a seed with an N gets
// KEY_INVALID (DnaString enumeration never yields N: index_iter.hpp:831).  The owning read is the
// last one whose scanned seed offset is <= the seed index: the proportional guess (exact for
// equal-length reads) is checked with loads that do not depend on each other, and only a wrong
// guess gallops / bisects.  Neighbouring threads read neighbouring bytes.
constexpr int SP = 1;        // seeds a thread works on at a time (more were measured slower: registers, occupancy)

// WIDE (seeds of 32..63 bases): the 128-bit k-mer goes to seed_wide, its fingerprint (table_key) to seed_key -- what the
// chunk's seed table is keyed by -- and its first pfx_len bases to seed_pfx (the prefix maps).
//
// PACKED (psigpu_find_seeds_packed): the reads arrive as 2-bit codes, 32 bases per u64 word, base i of the buffer in bits
// 63 - 2 (i % 32), 62 - 2 (i % 32) of word i / 32 -- first base most significant, so a k-mer is one funnel shift away from
// its key -- plus (optionally) one bit per base that says "not ACGT" (bit i % 64 of mask word i / 64).  `bases` is then
// the word array, `pk.bias2` / `pk.biasm` what to add to a base index of the call (read_off[r] + offset) to get its index
// in the word / mask buffers as they lie on the device (a sub-batch is transferred from a word boundary).
struct PackedIn { const uint64_t* mask; uint64_t bias2, biasm; };
// UNIFORM (PSIGPU_UNIFORM_READS): uni_len = the length every read is said to have, uni_spr = seeds per read; the owning
// read of seed s is s / uni_spr, and the claim is CHECKED for that read (a flag in the counters: the host then answers the
// chunk again the general way).
struct UniformIn { uint32_t len, spr; };

template <bool WIDE, bool PACKED = false, bool UNIFORM = false>
__global__ void __launch_bounds__(256)
k_seed_pack(const char* __restrict__ bases, const uint64_t* __restrict__ read_off,
            const uint64_t* __restrict__ seed_off, uint64_t n_reads, const uint64_t* __restrict__ params,
            uint64_t seeds_cap, uint64_t n_bases, uint32_t k, uint32_t step, uint64_t* __restrict__ seed_key, uint2* __restrict__ seed_info,
            DevCounters* ctr, u128* __restrict__ seed_wide, uint32_t* __restrict__ seed_pfx, uint32_t pfx_len, PackedIn pk = PackedIn{ nullptr, 0, 0 },
            UniformIn un = UniformIn{ 0, 0 })
{
  typedef typename std::conditional<WIDE, u128, uint64_t>::type KEY;
  constexpr uint32_t NW = WIDE ? 8 : 4;
  uint32_t nok = 0;
  // (UNIFORM: the seed count is the launch's own -- params[0] is written by a kernel in front of this one all the same)
  const uint64_t n_seeds = min(params[0], seeds_cap), ratio = params[1];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t nw = (k + 7) >> 3;               // 64-bit loads per seed (at most 4; 8 for two-word seeds)
  for (uint64_t s0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < n_seeds; s0 += stride * SP) {
    uint64_t lo[SP], so0[SP], so1[SP], ro[SP];
    bool in[SP];
    if constexpr (UNIFORM) {
      static_assert(SP == 1, "one seed per thread");
      const uint32_t r = (uint32_t)s0 / un.spr;              // (n_seeds < 2^32)
      in[0] = true; lo[0] = r; so0[0] = (uint64_t)r * un.spr; so1[0] = so0[0] + un.spr;
      ro[0] = (uint64_t)r * un.len;
      // the claim, checked where it is used: this read starts and ends where equal lengths put it
      if (read_off[r] != ro[0] || read_off[r + 1] != ro[0] + un.len) ctr->not_uniform.v = 1ull;
    } else {
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      const uint64_t s = s0 + (uint64_t)j * stride;
      in[j] = s < n_seeds;
      lo[j] = __umul64hi(s, ratio);
      if (lo[j] >= n_reads) lo[j] = n_reads - 1;
      so0[j] = 0; so1[j] = ~0ull; ro[j] = 0;
      if (in[j]) { so0[j] = seed_off[lo[j]]; so1[j] = seed_off[lo[j] + 1]; ro[j] = read_off[lo[j]]; }
    }
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      const uint64_t s = s0 + (uint64_t)j * stride;
      if (in[j] && !(so0[j] <= s && s < so1[j])) {
        // wrong guess (ragged reads): gallop to a bracket, bisect
        uint64_t l = lo[j], hi;
        if (so0[j] <= s) {
          uint64_t d = 1;
          while (l + d < n_reads && seed_off[l + d] <= s) { l += d; d <<= 1; }
          hi = min(l + d, n_reads);
        } else {
          uint64_t d = 1;
          hi = l;
          while (d < hi && seed_off[hi - d] > s) { hi -= d; d <<= 1; }
          l = d < hi ? hi - d : 0;
        }
        while (hi - l > 1) {                      // invariant: seed_off[l] <= s < seed_off[hi]
          uint64_t mid = (l + hi) >> 1;
          if (seed_off[mid] <= s) l = mid; else hi = mid;
        }
        lo[j] = l; so0[j] = seed_off[l]; ro[j] = read_off[l];
      }
    }
    }
    uint64_t x[SP][NW];
    uint64_t st[SP];
    bool fast[SP];
    if constexpr (PACKED) {
      static_assert(SP == 1, "one seed per thread");
      if (!in[0]) continue;
      const uint64_t s = s0;
      st[0] = (s - so0[0]) * step;
      const uint64_t* __restrict__ P = reinterpret_cast<const uint64_t*>(bases);
      const uint64_t q = ro[0] + st[0] + pk.bias2;
      const uint64_t w = q >> 5;
      const uint32_t sh = 2u * (uint32_t)(q & 31);
      // (the buffer is padded: the words behind the window are loaded but none of their bits is used)
      const uint64_t w0 = P[w], w1 = P[w + 1];
      const uint64_t hi = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
      KEY key;
      if constexpr (WIDE) {
        const uint64_t w2 = P[w + 2];
        const uint64_t lo2 = sh ? (w1 << sh) | (w2 >> (64 - sh)) : w1;
        key = (((u128)hi << 64) | (u128)lo2) >> (128 - 2 * k);
      } else key = hi >> (64 - 2 * k);
      uint32_t ok = 1;
      if (pk.mask) {
        const uint64_t qm = ro[0] + st[0] + pk.biasm;
        const uint32_t ms = (uint32_t)(qm & 63);
        const uint64_t m0 = pk.mask[qm >> 6], m1 = pk.mask[(qm >> 6) + 1];
        const uint64_t win = ms ? (m0 >> ms) | (m1 << (64 - ms)) : m0;
        ok = (win & ((k < 64 ? (1ull << k) : 0ull) - 1ull)) == 0;
      }
      if constexpr (WIDE) {
        seed_wide[s] = ok ? key : key_invalid<u128>();
        seed_pfx[s] = (uint32_t)(key >> (2 * (k - pfx_len)));
        seed_key[s] = ok ? table_key(key) : KEY_INVALID;
      } else
      seed_key[s] = ok ? key : KEY_INVALID;
      if (!UNIFORM || seed_info) seed_info[s] = make_uint2((uint32_t)lo[0], (uint32_t)st[0]);
      nok += ok;
      continue;
    }
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      const uint64_t s = s0 + (uint64_t)j * stride;
      st[j] = (s - so0[j]) * step;
      const uint64_t abs0 = ro[j] + st[j];
      fast[j] = in[j] && abs0 + 8ull * nw <= n_bases;
#pragma unroll
      for (uint32_t w = 0; w < NW; ++w) {
        x[j][w] = 0;
        if (fast[j] && w < nw) __builtin_memcpy(&x[j][w], bases + abs0 + 8 * w, 8);
      }
    }
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      if (!in[j]) continue;
      const uint64_t s = s0 + (uint64_t)j * stride;
      KEY key = 0;
      uint32_t ok = 1;
      if (fast[j]) {
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w)
          if (w < nw) {
            uint32_t take = min(8u, k - 8 * w);
            key = (key << (2 * take)) | (KEY)pack8(x[j][w], take, ok);
          }
      } else {
        const char* p = bases + ro[j] + st[j];
        for (uint32_t i = 0; i < k; ++i) {                // tail of the buffer: byte loads
          int b = base2(p[i]);
          if (b < 0) { ok = 0; b = 0; }
          key = (key << 2) | (KEY)b;
        }
      }
      if constexpr (WIDE) {
        seed_wide[s] = ok ? key : key_invalid<u128>();
        seed_pfx[s] = (uint32_t)(key >> (2 * (k - pfx_len)));
        seed_key[s] = ok ? table_key(key) : KEY_INVALID;
      } else
      seed_key[s] = ok ? key : KEY_INVALID;
      // (UNIFORM, answered from the k-mer table alone: nobody reads it -- the emit kernel derives both from the seed's number)
      if (!UNIFORM || seed_info) seed_info[s] = make_uint2((uint32_t)lo[j], (uint32_t)st[j]);     // (read, offset in read)
      nok += ok;
    }
  }
  // one atomic per wave
  for (int d = 32; d > 0; d >>= 1) nok += __shfl_down(nok, d);
  if (lane_id() == 0 && nok) ctr->n_seeds_valid.add((unsigned long long)nok);
}

// seeds "index" (the depth-k level of the reference's reads index, seed_finder.hpp:1089-1097, and the levels
// above it as prefix bitmaps), built per chunk for the query-time traverser.
//
// Rounds 1-2 built it with one device-scope CAS per seed into a table of the whole chunk plus one device-scope OR
// into a 4^14-bit map: 14 M random atomics, 1.0 ms per 7 M seeds (atomics retire at ~13 G/s on this part, loads at
// ~47 G/s).  Now the seeds are first PARTITIONED by their leading SB_BASES bases (count / scan / scatter: streams),
// and one workgroup per bucket builds the bucket's share of everything in LDS -- its slots of the table (the
// bucket's region: two slots per seed), its 4^(14-6) bits of the 14-mer map and its 4^(12-6) bits of the 12-mer map
// -- and writes them out whole: no global atomics, no separate reset of the table and the maps, no derive pass.
// A bucket too large for LDS (skewed sequence: poly-A prefixes) builds its region in place with atomics; nobody
// else touches that region.
// ------------------------------------------------------------------------------------
constexpr uint32_t SB_BASES = 6;                 // partition by this many leading bases (fewer when k is shorter)
constexpr uint32_t SB_TILE = 16384;              // seeds per workgroup in the count / scatter kernels
constexpr uint32_t SB_LDS_SLOTS = 4096;          // table slots a bucket may have to be built in LDS (64 KB: two workgroups per CU)
constexpr uint64_t SB_MAX_SEEDS = 9ull << 20;    // chunks up to this many seeds (upper bound) are partitioned: about 2000 seeds per bucket,
                                                 // 2048 fit the LDS table (a fuller bucket is built in place); larger chunks take the
                                                 // one-region build below

struct SeedBuckets {
  uint32_t pb;               // bases that select the bucket = min(SB_BASES, pfx_len)
  uint32_t n_buckets;        // 4^pb
  uint32_t n_wg;             // workgroups of the count / scatter kernels
  uint32_t k;
};

__device__ __forceinline__ uint32_t sb_bucket(uint64_t key, uint32_t k, uint32_t pb) { return pb ? (uint32_t)(key >> (2 * (k - pb))) : 0u; }
// where a k-mer's search starts inside its bucket's region of m slots
__device__ __forceinline__ uint32_t sb_home(uint64_t key, uint32_t m) { return (uint32_t)__umul64hi(mix64(key), (uint64_t)m); }

__global__ void __launch_bounds__(256)
k_sb_count(const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params, uint64_t seeds_cap, SeedBuckets sb,
           uint32_t* __restrict__ cnt /* [bucket][wg] */)
{
  extern __shared__ uint32_t hist[];
  for (uint32_t i = threadIdx.x; i < sb.n_buckets; i += 256) hist[i] = 0;
  __syncthreads();
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = (uint64_t)blockIdx.x * SB_TILE, s1 = min(n_seeds, s0 + SB_TILE);
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += 256 * 8) {        // eight independent loads per thread in flight
    uint64_t key[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) key[j] = s + 256 * j < s1 ? seed_key[s + 256 * j] : KEY_INVALID;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (key[j] != KEY_INVALID) atomicAdd(&hist[sb_bucket(key[j], sb.k, sb.pb)], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < sb.n_buckets; i += 256) cnt[(uint64_t)i * sb.n_wg + blockIdx.x] = hist[i];
}

__global__ void __launch_bounds__(256)
k_sb_scatter(const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params, uint64_t seeds_cap, SeedBuckets sb,
             const uint64_t* __restrict__ off /* exclusive scan of cnt */, ulonglong2* __restrict__ out_rec /* (k-mer, seed) */,
             uint32_t* __restrict__ seed_next)
{
  extern __shared__ uint32_t cur[];
  for (uint32_t i = threadIdx.x; i < sb.n_buckets; i += 256) cur[i] = (uint32_t)off[(uint64_t)i * sb.n_wg + blockIdx.x];
  __syncthreads();
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = (uint64_t)blockIdx.x * SB_TILE, s1 = min(n_seeds, s0 + SB_TILE);
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += 256 * 8) {
    uint64_t key[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) key[j] = s + 256 * j < s1 ? seed_key[s + 256 * j] : KEY_INVALID;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s + 256 * j < s1) seed_next[s + 256 * j] = NIL;
      if (key[j] == KEY_INVALID) continue;
      const uint32_t at = atomicAdd(&cur[sb_bucket(key[j], sb.k, sb.pb)], 1u);
      // one 16-byte store per seed: the stores of a bucket's run come from all over the tile and leave L2 as partial
      // lines, one request each -- k-mer and seed number in two arrays were two requests (0.19 -> 0.1x ms)
      out_rec[at] = make_ulonglong2(key[j], s + 256 * j);
    }
  }
}

// one workgroup per bucket
__global__ void __launch_bounds__(256)
k_sb_build(const ulonglong2* __restrict__ brec, const uint64_t* __restrict__ off, SeedBuckets sb,
           TableSlot* __restrict__ ht, uint32_t* __restrict__ seed_next, uint32_t* __restrict__ pfx_bits, uint32_t pfx_len,
           uint32_t* __restrict__ pfx12)
{
  __shared__ TableSlot tab[SB_LDS_SLOTS];
  __shared__ uint32_t bm[2048];                   // 4^(14 - 6) bits at most
  const uint32_t b = blockIdx.x;
  const uint32_t lo = (uint32_t)off[(uint64_t)b * sb.n_wg], hi = (uint32_t)off[(uint64_t)(b + 1) * sb.n_wg];
  const uint32_t n = hi - lo, m = 2 * n;          // the bucket's region: slots [2 lo, 2 hi)
  const uint32_t sub = pfx_len - sb.pb;           // bases of the 14-mer prefix inside the bucket
  const uint32_t bm_words = ((1u << (2 * sub)) + 31) / 32;
  for (uint32_t i = threadIdx.x; i < bm_words; i += 256) bm[i] = 0;
  const bool in_lds = m <= SB_LDS_SLOTS;
  TableSlot* region = ht + 2ull * lo;
  const TableSlot empty = { KEY_INVALID, NIL, NIL };
  if (in_lds) { for (uint32_t i = threadIdx.x; i < m; i += 256) tab[i] = empty; }
  else { for (uint32_t i = threadIdx.x; i < m; i += 256) region[i] = empty; __threadfence(); }
  __syncthreads();
  TableSlot* t = in_lds ? tab : region;
  const uint32_t sh = 2 * (sb.k - pfx_len);
  const uint32_t sub_mask = (1u << (2 * sub)) - 1u;
  auto insert = [&](const ulonglong2 rec) {
    const uint64_t key = rec.x;
    const uint32_t s = (uint32_t)rec.y;
    const uint32_t pf = (uint32_t)(key >> sh) & sub_mask;
    atomicOr(&bm[pf >> 5], 1u << (pf & 31));
    uint32_t h = sb_home(key, m);
    while (true) {
      unsigned long long prev = atomicCAS(&t[h].key, (unsigned long long)KEY_INVALID, (unsigned long long)key);
      if (prev == KEY_INVALID) { t[h].val = s; break; }
      if (prev == key) { seed_next[s] = atomicExch(&t[h].dup, s); break; }
      h = h + 1 < m ? h + 1 : 0;
    }
  };
  if (in_lds) {
    // a bucket that fits LDS has at most 2048 records: eight per thread, all requested before the first is inserted (one
    // memory latency per workgroup instead of one per record: the loop of loads behind atomics was 7 latencies deep and the
    // kernel, two workgroups per CU, waited for them: 0.154 -> 0.1x ms)
    ulonglong2 r[SB_LDS_SLOTS / 2 / 256];
#pragma unroll
    for (uint32_t j = 0; j < SB_LDS_SLOTS / 2 / 256; ++j) {
      const uint32_t i = threadIdx.x + 256 * j;
      r[j] = i < n ? brec[lo + i] : make_ulonglong2(KEY_INVALID, 0);
    }
#pragma unroll
    for (uint32_t j = 0; j < SB_LDS_SLOTS / 2 / 256; ++j)
      if (threadIdx.x + 256 * j < n) insert(r[j]);
  } else
  for (uint32_t i = threadIdx.x; i < n; i += 256) insert(brec[lo + i]);
  __syncthreads();
  if (in_lds) {
    const uint4* src = reinterpret_cast<const uint4*>(tab);
    uint4* dst = reinterpret_cast<uint4*>(region);
    for (uint32_t i = threadIdx.x; i < m; i += 256) dst[i] = src[i];
  }
  // the bucket's bits of the seed-prefix maps: bucket b owns bits [b 4^sub, (b + 1) 4^sub) of the 4^pfx_len-bit map
  if (bm_words * 32 == (1u << (2 * sub))) {
    for (uint32_t i = threadIdx.x; i < bm_words; i += 256) pfx_bits[(uint64_t)b * bm_words + i] = bm[i];
  } else {                                        // fewer than 32 bits per bucket (short prefixes): shared words
    if (threadIdx.x == 0 && bm[0]) atomicOr(&pfx_bits[((uint64_t)b << (2 * sub)) >> 5], bm[0] << (((uint64_t)b << (2 * sub)) & 31));
  }
  if (pfx12 != nullptr) {
    // a 12-mer is a seed prefix iff one of its 4^(pfx_len - 12) extensions is; this bucket owns 4^(12 - pb) of them
    const uint32_t ext = 1u << (2 * (pfx_len - PFX_SHORT));            // 16 (pfx_len 14) or 4 (13)
    const uint32_t n12 = 1u << (2 * (PFX_SHORT - sb.pb));               // 4096 at pb = 6
    for (uint32_t w = threadIdx.x; w < n12 / 32; w += 256) {
      uint32_t o = 0;
      for (uint32_t j = 0; j < 32; ++j) {
        const uint32_t first = (w * 32 + j) * ext;                      // first bit of the group in bm
        const uint32_t g = (bm[first >> 5] >> (first & 31)) & (ext == 16 ? 0xFFFFu : 0xFu);
        o |= (g ? 1u : 0u) << j;
      }
      pfx12[(uint64_t)b * (n12 / 32) + w] = o;
    }
  }
}

// The same table for chunks whose buckets would not fit LDS (more than SB_MAX_SEEDS seeds: a bucket of the partition
// above would hold more than SB_LDS_SLOTS / 2, and a finer partition would need a count matrix larger than the
// data): ONE region for all seeds, reset by k_fill3, one CAS per seed plus one OR into the 4^pfx_len-bit map, the
// 12-mer map derived afterwards -- the build of rounds 1-2.  Same slots, same addressing (sb_home over the region),
// so the traverser's lookup does not know the difference (one bucket: pb = 0).
struct FillJob { uint4* p; uint64_t n16; uint32_t v; };

__global__ void __launch_bounds__(256) k_fill3(FillJob a, FillJob b, FillJob c)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (uint64_t i = t0; i < a.n16; i += stride) a.p[i] = make_uint4(a.v, a.v, a.v, a.v);
  for (uint64_t i = t0; i < b.n16; i += stride) b.p[i] = make_uint4(b.v, b.v, b.v, b.v);
  for (uint64_t i = t0; i < c.n16; i += stride) c.p[i] = make_uint4(c.v, c.v, c.v, c.v);
}

__global__ void k_table_insert(const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params,
                               uint64_t seeds_cap, TableSlot* __restrict__ ht, uint32_t m /* slots of the one region */,
                               uint32_t* __restrict__ seed_next, uint32_t k,
                               uint32_t* __restrict__ pfx_bits, uint32_t pfx_len, uint64_t* __restrict__ boff,
                               const uint32_t* __restrict__ seed_pfx /* two-word seeds: the key is a fingerprint, the prefix comes from here */)
{
  uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) { boff[0] = 0; boff[1] = m / 2; }       // the one bucket's "offsets": slots [2 boff[0], 2 boff[1])
  if (s >= min(params[0], seeds_cap)) return;
  uint64_t key = seed_key[s];
  seed_next[s] = NIL;
  if (key == KEY_INVALID) return;
  {
    uint32_t pf = seed_pfx ? seed_pfx[s] : (uint32_t)(key >> (2 * (k - pfx_len)));
    atomicOr(&pfx_bits[pf >> 5], 1u << (pf & 31));
  }
  uint32_t h = sb_home(key, m);
  while (true) {
    unsigned long long prev = atomicCAS(&ht[h].key, (unsigned long long)KEY_INVALID, (unsigned long long)key);
    if (prev == KEY_INVALID) { ht[h].val = (uint32_t)s; return; }
    if (prev == key) { seed_next[s] = atomicExch(&ht[h].dup, (uint32_t)s); return; }
    h = h + 1 < m ? h + 1 : 0;
  }
}

// first-level bitmap (4^12 bits) derived from the second level: a 12-mer is a seed prefix iff
// one of its 4^(pfx_len-12) extensions is.  One thread per output word, no atomics.
__global__ void k_pfx_derive(const uint32_t* __restrict__ pfx_bits, uint32_t pfx_len,
                             uint32_t* __restrict__ pfx12)
{
  uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;            // output word: 32 12-mers
  if (w >= (1u << (2 * PFX_SHORT)) / 32) return;
  uint32_t ext = 1u << (2 * (pfx_len - PFX_SHORT));               // bits per 12-mer in the source: 4 or 16
  uint32_t out = 0;
  if (ext == 16) {
    const uint4* src = reinterpret_cast<const uint4*>(pfx_bits + (uint64_t)w * 16);
    for (int i = 0; i < 4; ++i) {
      uint4 v = src[i];
      uint32_t x[4] = { v.x, v.y, v.z, v.w };
      for (int j = 0; j < 4; ++j) {
        out |= ((x[j] & 0xFFFFu) ? 1u : 0u) << (8 * i + 2 * j);
        out |= ((x[j] >> 16) ? 1u : 0u) << (8 * i + 2 * j + 1);
      }
    }
  } else {                                                        // ext == 4 (pfx_len 13)
    const uint4* src = reinterpret_cast<const uint4*>(pfx_bits + (uint64_t)w * 4);
    uint4 v = src[0];
    uint32_t x[4] = { v.x, v.y, v.z, v.w };
    for (int j = 0; j < 4; ++j)
      for (int b = 0; b < 8; ++b) out |= (((x[j] >> (4 * b)) & 0xFu) ? 1u : 0u) << (8 * j + b);
  }
  pfx12[w] = out;
}

// ------------------------------------------------------------------------------------
// K1: FM backward search, one quad per seed.  The text is the FORWARD path text, so the
// seed is consumed from its last base to its first (the reference appends characters to a
// pattern on the REVERSED text: index_iter.hpp:820-824 -- same occurrences).
// Interval [l, r) half-open; l' = C[c] + rank_c(l), r' = C[c] + rank_c(r)
// (sdsl::backward_search behind fmindex.hpp:856).
// ------------------------------------------------------------------------------------
// Every wave owns a contiguous range of `per_wave` seeds (a multiple of 16) and walks it 16
// seeds -- one per quad -- at a time.  It leaves (lo, count) per seed and the sum of its counts;
// k_wave_offsets turns the sums into the wave's first output slot, and k_fm_locate, walking the
// same ranges, places every hit with a running wave-local prefix: hits come out in seed order
// with no atomics and no scan over the seeds.
template <bool LISTED, typename KEY>   // (LISTED: two kernels -- the list mode's pointers and strides cost the range mode scalar
                              // registers, and at 101 of them a SIMD holds 7 waves instead of the 8 the launch is sized
                              // for.  KEY: one word for seeds of up to 31 bases, two for up to 63)
__global__ void __launch_bounds__(256)
k_fm_search(FMView fm, const KEY* __restrict__ seed_key, const uint64_t* __restrict__ params,
            uint64_t seeds_cap, uint32_t per_wave,
            uint32_t k, uint32_t gocc_thr, uint32_t* __restrict__ iv_lo, uint32_t* __restrict__ iv_cnt,
            uint32_t* __restrict__ iv_aux, uint64_t* __restrict__ wave_total, DevCounters* ctr,
            const uint32_t* __restrict__ list, const unsigned long long* __restrict__ n_list)
{
  // Two ways to be given work: every wave owns the contiguous seed range [wave * per_wave, ...)
  // (list == nullptr), or the waves share a list of seed indices -- the seeds k_fm_search_direct
  // deferred -- and add each seed's count to the total of the wave that owns its range.
  __shared__ uint32_t s_sup[SUP_LDS];
  stage_exc_super(fm, s_sup);
  const bool can_verify = fm.text4 != nullptr && fm.sa != nullptr;
  constexpr bool listed = LISTED;
  const uint32_t ql = threadIdx.x & 3, quad = (threadIdx.x & 63) >> 2;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = listed ? wave * 16 : wave * per_wave;
  const uint64_t s1 = listed ? min((uint64_t)*n_list, seeds_cap) : min(n_seeds, s0 + per_wave);
  const uint64_t stride = listed ? ((uint64_t)gridDim.x * blockDim.x >> 6) * 16 : 16;
  uint32_t n_live = 0, n_steps = 0, n_rows = 0;
  uint64_t wsum = 0;
  for (uint64_t base = s0; base < s1; base += stride) {
    const bool in = base + quad < s1;
    const uint64_t seed = listed ? (in ? list[base + quad] : 0) : base + quad;
    KEY key = in ? seed_key[seed] : key_invalid<KEY>();
    bool alive = key != key_invalid<KEY>();
    uint32_t l = 0, r = fm.n, j0 = 0;
    if (fm.ftab != nullptr && k >= fm.ftab_len) {
      // the first ftab_len steps (the seed's last ftab_len bases) are one table lookup
      j0 = fm.ftab_len;
      if (alive) {
        uint2 iv = fm.ftab[(uint64_t)key & ((1ull << (2 * j0)) - 1ull)];
        l = iv.x; r = iv.y;
        alive = r > l;
      }
    }
    // LF steps, per quad, until the seed is exhausted -- or until its interval is small and the
    // rest of the seed short enough to be checked against the text itself (whole SA resident)
    uint32_t jq = j0;
    while (true) {
      bool step = alive && jq < k && !(can_verify && (r - l) <= VERIFY_ROWS && (k - jq) <= 16u);
      if (!__any(step)) break;
      if (step) {
        uint32_t c = (uint32_t)(key >> (2 * jq)) & 3u;
        uint32_t bl = l / BLOCK_SYMS, br = r / BLOCK_SYMS;
        uint4 vl = fm.blocks[(uint64_t)bl * 4 + ql];
        uint4 vr = vl;
        if (br != bl) vr = fm.blocks[(uint64_t)br * 4 + ql];
        uint32_t nl = fm.C[c] + quad_rank(fm, s_sup, vl, ql, c, l);
        uint32_t nr = fm.C[c] + quad_rank(fm, s_sup, vr, ql, c, r);
        l = nl; r = nr;
        alive = r > l;
        ++jq;
        n_steps += ql == 0;
      }
    }
    // verification: the quad's lanes take the interval's rows four at a time
    uint32_t cnt = alive ? r - l : 0u, aux = 0;
    if (__any(alive && jq < k)) {
      uint32_t rem = k - jq, mask = 0;
      if (alive && jq < k) {
        for (uint32_t t = ql; t < r - l; t += 4)
          if (text_matches(fm.text4, fm.sa[l + t], rem, key, k)) mask |= 1u << t;
      }
      mask = quad_sum(mask);                       // disjoint bits: sum == or
      if (alive && jq < k) { n_rows += ql == 0 ? r - l : 0u; cnt = (uint32_t)__popc(mask); aux = (rem << 8) | mask; }
    }
    // seeds above the gocc threshold are dropped here (index_iter.hpp:843-847)
    bool keep = cnt != 0 && cnt <= gocc_thr;
    if (in && ql == 0) {
      iv_lo[seed] = l;
      iv_cnt[seed] = keep ? cnt : 0u;
      iv_aux[seed] = aux;
      n_live += keep;
      wsum += keep ? cnt : 0u;
      if (listed && keep) atomicAdd((unsigned long long*)&wave_total[seed / per_wave], (unsigned long long)cnt);
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_live += __shfl_down(n_live, d); wsum += __shfl_down(wsum, d);
    n_steps += __shfl_down(n_steps, d); n_rows += __shfl_down(n_rows, d);
  }
  if (lane_id() == 0) {
    if (!listed) wave_total[wave] = wsum;
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
    if (n_steps) ctr->n_lf_steps.add((unsigned long long)n_steps);
    if (n_rows) ctr->n_rows_verified.add((unsigned long long)n_rows);
  }
}

// exclusive scan of the per-wave totals (at most WAVES_MAX values): one workgroup of 1024 threads,
// WAVES_MAX / 1024 values per thread, wave shuffles + one LDS hop
constexpr int WAVES_MAX = 8192;       // waves of K1 / the probe / K2: all resident at once (16 K and 32 K measured slower)

__global__ void __launch_bounds__(1024)
k_wave_offsets(uint64_t* wave_total, const uint64_t* __restrict__ wave_total_off, uint64_t n_waves, uint64_t* total_on,
               uint64_t* total_all, bool accumulate)
{
  // accumulate: this is a further part of the index -- its hits go behind those already counted in *total_all.
  // wave_total[n_waves] / [n_waves + 1] get the part's range of output slots.
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t osum[16];
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  constexpr int V = WAVES_MAX / 1024;
  const uint64_t base = accumulate ? *total_all : 0, base_on = accumulate ? *total_on : 0;
  uint64_t v[V], s = 0, on = 0;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    uint64_t idx = (uint64_t)t * V + i;
    uint64_t a = idx < n_waves ? wave_total[idx] : 0;
    on += a;
    v[i] = a + ((wave_total_off && idx < n_waves) ? wave_total_off[idx] : 0);
    s += v[i];
  }
  for (int d = 32; d > 0; d >>= 1) on += __shfl_down(on, d);
  if (lane == 0) osum[w] = on;
  uint64_t incl = s;
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t u = __shfl_up(incl, d);
    if (lane >= (uint32_t)d) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint64_t before = 0, all = 0;
  for (uint32_t i = 0; i < 16; ++i) { if (i < w) before += wsum[i]; all += wsum[i]; }
  uint64_t run = base + before + incl - s;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    uint64_t idx = (uint64_t)t * V + i;
    if (idx < n_waves) wave_total[idx] = run;
    run += v[i];
  }
  if (t == 0) {
    uint64_t o = 0;
    for (uint32_t i = 0; i < 16; ++i) o += osum[i];
    *total_on = base_on + o;
    *total_all = base + all;
    wave_total[n_waves] = base;
    wave_total[n_waves + 1] = base + all;
  }
}

// An index in several parts: K1 ran once per part and left every part's occurrence count per seed.  A gocc
// threshold counts a k-mer's occurrences in the whole path text (index_iter.hpp:843-847): seeds whose counts add
// up to more than `thr` lose them in every part; the per-wave totals of every part are made here (K1's own were
// taken before the threshold), and the seeds with an occurrence in any part are counted once.
__global__ void __launch_bounds__(256)
k_parts_combine(uint32_t* __restrict__ iv_cnt, uint64_t seed_stride, uint32_t n_parts, uint32_t thr,
                const uint64_t* __restrict__ params, uint64_t seeds_cap, uint32_t per_wave,
                uint64_t* __restrict__ wave_total, uint64_t tiles_stride, DevCounters* ctr)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  uint64_t wsum[PSIGPU_MAX_PARTS];
  uint32_t n_live = 0;
  for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] = 0;
  for (uint64_t seed = s0 + lane; seed < s1; seed += 64) {
    uint64_t total = 0;
    uint32_t c[PSIGPU_MAX_PARTS];
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) { c[p] = p < n_parts ? iv_cnt[p * seed_stride + seed] : 0u; total += c[p]; }
    const bool keep = total != 0 && total <= thr;
    if (total != 0 && !keep)
      for (uint32_t p = 0; p < n_parts; ++p) if (c[p]) iv_cnt[p * seed_stride + seed] = 0;
    n_live += keep;
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] += keep ? c[p] : 0u;
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_live += __shfl_down(n_live, d);
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] += __shfl_down(wsum[p], d);
  }
  if (lane == 0) {
    for (uint32_t p = 0; p < n_parts; ++p) wave_total[p * tiles_stride + wave] = wsum[p];
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
  }
}

// ------------------------------------------------------------------------------------
// K2: locate + map + emit.  What every K2 variant reads of the index (k_fm_locate_direct and k_kmer_emit
// when the whole suffix array is resident, k_fm_walk + k_hits_resolve when it is sampled).
// ------------------------------------------------------------------------------------
struct MapView {
  const uint32_t* samples; uint32_t sa_rate;
  const uint32_t* exc_sa;
  const SegRec* seg;            // n_segs + 1 records (the last one is a sentinel at text_len)
  const uint32_t* seg_dir;
  const SaRec* sarec;           // per-row records for seeds with sarec_rem bases in front of the q-mer, or nullptr
  uint32_t sarec_rem;
  const uint2* saloc;           // (node rank, offset) of SA[row] for every row, or nullptr
  const uint2* loci;            // starting loci (node rank, offset): what the tables' locus runs index
  const uint2* on_pos;          // k-mer table: (node rank, offset) runs of the path k-mers with several occurrences
  const uint64_t* node_id;      // rank -> external id ...
  uint64_t id_base;             // ... or id = rank + id_base when the ids are consecutive
  bool id_affine;
};

// K5 (emission).  No per-hit atomics anywhere:
//  * on-path hits are placed by an exclusive scan over the per-seed interval sizes: seed i
//    owns hits[off_i, off_i + cnt_i) -- deterministic, in seed order;
//  * the traverser writes into private 256-record chunks (one atomic per chunk to take the
//    next one), records how full each chunk got, and k_chunk_compact packs the chunks behind
//    the on-path hits.
constexpr uint32_t CHUNK = 256;           // records per traverser output chunk (8 KB)

struct ChunkWriter {
  psigpu_hit* chunks;        // cap_chunks x CHUNK records
  uint32_t* fill;            // [cap_chunks], zero-initialised
  uint32_t cap_chunks;
  uint32_t id;               // wave-uniform: current chunk, 0xFFFFFFFF = none / overflowed
  uint32_t n;                // wave-uniform: records in the current chunk
};

// wave-uniform control flow required
__device__ __forceinline__ void chunk_emit(ChunkWriter& w, bool has, uint64_t node_id, uint64_t noff,
                                           uint64_t rid, uint64_t roff, DevCounters* ctr)
{
  uint64_t m = __ballot(has);
  if (m == 0) return;
  uint32_t add = (uint32_t)__popcll(m);
  if (w.id == NIL || w.n + add > CHUNK) {
    if (w.id != NIL && w.id < w.cap_chunks && lane_id() == 0) w.fill[w.id] = w.n;
    unsigned long long nid = 0;
    if (lane_id() == 0) nid = atomicAdd(&ctr->n_chunks.v, 1ull);
    w.id = (uint32_t)__shfl(nid, 0);
    w.n = 0;
  }
  if (has && w.id < w.cap_chunks) {
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(
        w.chunks + (uint64_t)w.id * CHUNK + w.n + (uint32_t)__popcll(m & lanemask_lt()));
    dst[0] = make_ulonglong2(node_id, noff);
    dst[1] = make_ulonglong2(rid, roff);
  }
  w.n += add;
}

__device__ __forceinline__ void chunk_close(ChunkWriter& w)
{
  if (w.id != NIL && w.id < w.cap_chunks && lane_id() == 0) w.fill[w.id] = w.n;
}

// Enumeration mode of the traverser (building the locus k-mer table): completed walks leave
// (k-mer, locus) pairs in 16-byte records, same private-chunk scheme.
struct EnumOut {
  ulonglong2* chunks;        // cap_chunks x CHUNK pairs
  uint32_t* fill;
  uint32_t cap_chunks;
  uint32_t* walks;           // [n_loci] complete walks seen per locus
  uint32_t walk_cap;         // loci with more walks than this stay with the query-time traverser
  uint32_t prefix;           // 1: the walks are the loci's PREFIX walks (ensure_pfx_roots): every pair also carries where the
                             // walk stands after its last base -- (node | k-mer, offset | locus) -- so that it can be resumed
};

struct PairWriter { uint32_t id, n; };

__device__ __forceinline__ void pair_emit(const EnumOut& eo, PairWriter& w, bool has, uint64_t kmer, uint64_t locus,
                                          DevCounters* ctr)
{
  uint64_t m = __ballot(has);
  if (m == 0) return;
  uint32_t add = (uint32_t)__popcll(m);
  if (w.id == NIL || w.n + add > CHUNK) {
    if (w.id != NIL && w.id < eo.cap_chunks && lane_id() == 0) eo.fill[w.id] = w.n;
    unsigned long long nid = 0;
    if (lane_id() == 0) nid = atomicAdd(&ctr->n_chunks.v, 1ull);
    w.id = (uint32_t)__shfl(nid, 0);
    w.n = 0;
  }
  if (has && w.id < eo.cap_chunks)
    eo.chunks[(uint64_t)w.id * CHUNK + w.n + (uint32_t)__popcll(m & lanemask_lt())] = make_ulonglong2(kmer, locus);
  w.n += add;
}

// one workgroup per chunk: copy its records behind the on-path hits
__global__ void __launch_bounds__(256)
k_chunk_compact(const psigpu_hit* __restrict__ chunks, const uint32_t* __restrict__ fill,
                const uint64_t* __restrict__ chunk_off, uint32_t cap_chunks,
                const unsigned long long* __restrict__ n_on, psigpu_hit* __restrict__ hits, uint64_t cap)
{
  uint32_t c = blockIdx.x;
  if (c >= cap_chunks) return;
  uint32_t n = fill[c];
  uint64_t dst0 = *n_on + chunk_off[c];
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(chunks + (uint64_t)c * CHUNK);
  for (uint32_t i = threadIdx.x; i < 2 * n; i += blockDim.x) {
    uint64_t rec = dst0 + (i >> 1);
    if (rec < cap) reinterpret_cast<ulonglong2*>(hits + rec)[i & 1] = src[i];
  }
}


// ------------------------------------------------------------------------------------
// Locus k-mer table.  The starting loci and k are fixed when the index is made, so the walks the
// traverser would enumerate from them for every chunk (traverser_bfs.hpp:72-161) can be
// enumerated ONCE, when the index is loaded: k_traverse<ENUM> leaves every (k-mer, locus) pair,
// the pairs are sorted by k-mer, and a static open-addressing table maps each distinct k-mer to
// its run of loci.  A chunk's off-path hits are then one probe per seed (k_lkt_probe) and are
// emitted by K2 right behind the seed's on-path hits; per-chunk cost no longer depends on the
// number of loci.  Loci with more than walk_cap walks (dense, high-degree regions: the number of
// walks is exponential there) are left out and stay with the query-time traverser, which prunes
// them with the chunk's seeds.
// ------------------------------------------------------------------------------------
// Runs of loci (k-mers spelled from several starting loci): indices into the loci array, sorted
// by k-mer -- the value array of the sort itself, 4 bytes per k-walk.
typedef uint32_t LocusEnt;

struct LktView {
  const TableSlot* ht;       // key = k-mer, val = first entry, dup = number of entries
  uint64_t n_slots;          // any size (not a power of two: the whole-genome table has to fit): slot = hash * n / 2^64
  const LocusEnt* ent;
};
__device__ __forceinline__ uint64_t lkt_home(uint64_t key, uint64_t n_slots) { return __umul64hi(mix64(key), n_slots); }
__device__ __forceinline__ uint64_t lkt_next(uint64_t h, uint64_t n_slots) { return h + 1 < n_slots ? h + 1 : 0; }

// one workgroup per enumeration chunk: split the pairs into key / value arrays for the sort;
// pairs of loci over the walk cap get a key above every k-mer (they sort to the end)
__global__ void __launch_bounds__(256)
k_enum_compact(const ulonglong2* __restrict__ chunks, const uint32_t* __restrict__ fill,
               const uint64_t* __restrict__ chunk_off, uint32_t cap_chunks, const uint32_t* __restrict__ walks,
               uint32_t walk_cap, uint32_t k, const uint32_t* __restrict__ id_map, uint64_t* __restrict__ keys,
               uint32_t* __restrict__ vals, unsigned long long* __restrict__ n_dropped)
{
  uint32_t c = blockIdx.x;
  if (c >= cap_chunks) return;
  uint32_t n = fill[c];
  uint64_t dst0 = chunk_off[c];
  uint32_t dropped = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    ulonglong2 r = chunks[(uint64_t)c * CHUNK + i];
    uint32_t locus = (uint32_t)r.y;
    bool drop = walks[locus] > walk_cap;
    keys[dst0 + i] = drop ? (1ull << (2 * k)) : r.x;
    vals[dst0 + i] = id_map ? id_map[locus] : locus;      // second pass: index in the left-over list -> locus
    dropped += drop;
  }
  for (int d = 32; d > 0; d >>= 1) dropped += __shfl_down(dropped, d);
  if (lane_id() == 0 && dropped) atomicAdd(n_dropped, (unsigned long long)dropped);
}

// sorted pairs -> table: the first entry of every run of equal k-mers claims a slot
// prefix walks (ensure_pfx_roots): one workgroup per enumeration chunk -- (12-mer | node << 32, locus | offset << 32) pairs to
// (12-mer, node, offset, locus) records, and the sort key (the locus) of every record
__global__ void __launch_bounds__(256)
k_pfx_compact(const ulonglong2* __restrict__ chunks, const uint32_t* __restrict__ fill, const uint64_t* __restrict__ chunk_off,
              uint32_t cap_chunks, uint4* __restrict__ out, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
  const uint32_t c = blockIdx.x;
  if (c >= cap_chunks) return;
  const uint32_t n = fill[c];
  const uint64_t dst0 = chunk_off[c];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const ulonglong2 r = chunks[(uint64_t)c * CHUNK + i];
    out[dst0 + i] = make_uint4((uint32_t)r.x, (uint32_t)(r.x >> 32), (uint32_t)(r.y >> 32), (uint32_t)r.y);
    keys[dst0 + i] = (uint32_t)r.y;
    vals[dst0 + i] = (uint32_t)(dst0 + i);
  }
}

__global__ void __launch_bounds__(256)
k_pfx_gather(const uint4* __restrict__ in, const uint32_t* __restrict__ order, uint64_t n, uint4* __restrict__ out)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[order[i]];
}

__global__ void k_lkt_insert(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                             const uint2* __restrict__ loci, uint64_t n, TableSlot* __restrict__ ht, uint64_t n_slots)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = keys[i];
  if (i && keys[i - 1] == key) return;
  // run length: gallop, then bisect (runs are almost always 1 or 2 long)
  uint64_t lo = i, stepw = 1;                    // keys[lo] == key
  while (lo + stepw < n && keys[lo + stepw] == key) { lo += stepw; stepw <<= 1; }
  uint64_t hi = lo + stepw < n ? lo + stepw : n; // keys[hi] != key or hi == n
  while (hi - lo > 1) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  // a k-mer with a single locus keeps it in the slot: no second access at query time
  const bool single = hi - i == 1;
  uint2 lc = make_uint2(0, 0);
  if (single) lc = loci[vals[i]];
  uint64_t h = lkt_home(key, n_slots);
  while (true) {
    unsigned long long prev = atomicCAS(&ht[h].key, (unsigned long long)KEY_INVALID, (unsigned long long)key);
    if (prev == KEY_INVALID) {
      if (single) { ht[h].val = lc.x; ht[h].dup = lc.y; __threadfence(); ht[h].key = key | LKT_INLINE; }
      else { ht[h].val = (uint32_t)i; ht[h].dup = (uint32_t)(hi - i); }
      return;
    }
    h = lkt_next(h, n_slots);
  }
}

// loci over the walk cap, in locus order within a wave
__global__ void k_lkt_residual(const uint32_t* __restrict__ walks, uint64_t n_loci, uint32_t walk_cap,
                               const uint2* __restrict__ loci, const uint32_t* __restrict__ ids_in, uint2* __restrict__ out,
                               uint32_t* __restrict__ ids_out, unsigned long long* __restrict__ n_out)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool r = i < n_loci && walks[i] > walk_cap;
  uint64_t m = __ballot(r);
  if (m == 0) return;
  unsigned long long base = 0;
  if (lane_id() == 0) base = atomicAdd(n_out, (unsigned long long)__popcll(m));
  base = __shfl(base, 0);
  if (r && out) {
    const uint64_t at = base + __popcll(m & lanemask_lt());
    out[at] = loci[i];
    if (ids_out) ids_out[at] = ids_in ? ids_in[i] : (uint32_t)i;
  }
}

// resolve a probe whose first slot `sl` (at index h) has been loaded
__device__ __forceinline__ void lkt_resolve(const LktView& lk, uint64_t key, uint64_t h, TableSlot sl,
                                            uint32_t& first, uint32_t& cnt, uint32_t& noff)
{
  first = 0; cnt = 0; noff = 0;
  while (true) {
    if (sl.key == KEY_INVALID) return;
    if ((sl.key & ~LKT_INLINE) == key) {
      if (sl.key & LKT_INLINE) { first = sl.val; noff = sl.dup; cnt = 1u | OFF_INLINE; }
      else { first = sl.val; cnt = sl.dup; }
      return;
    }
    h = lkt_next(h, lk.n_slots);
    sl = lk.ht[h];
  }
}

// Query side when K1 does not carry the probe (no path index, or K1's quad kernel): one lane per
// seed, the wave ranges of K1 / K2.  Leaves the seed's run in the table and the wave's total.
__global__ void __launch_bounds__(256)
k_lkt_probe(LktView lk, const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params,
            uint64_t seeds_cap, uint32_t per_wave, SeedOut so, uint64_t* __restrict__ wave_total_off)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  uint64_t wsum = 0;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t seed = base + lane;
    if (seed >= s1) continue;
    uint64_t key = seed_key[seed];
    uint32_t first = 0, cnt = 0, noff = 0;
    if (key != KEY_INVALID) {
      uint64_t h = lkt_home(key, lk.n_slots);
      lkt_resolve(lk, key, h, lk.ht[h], first, cnt, noff);
    }
    so.off_first[seed] = first;
    so.off_cnt[seed] = cnt;
    so.off_noff[seed] = noff;
    wsum += cnt & ~OFF_INLINE;
  }
  for (int d = 32; d > 0; d >>= 1) wsum += __shfl_down(wsum, d);
  if (lane == 0) wave_total_off[wave] = wsum;
}

// ------------------------------------------------------------------------------------
// k-mer table (PSIGPU_MODE_KMER_TABLE).  With the seed length fixed by the index, a seed is a
// key: the table maps every k-mer of the indexed paths to its suffix-array interval (what the
// backward search of K1 would return) and every k-mer spelled by a k-walk from a starting locus
// to its run in the locus entries (what the traverser would find), in ONE slot -- a seed costs
// one random sector instead of interval table + row records + locus table.  A k-mer with a
// single occurrence / a single locus carries that position in the slot.  The FM-index kernels
// stay the path for any other seed length and when the table does not fit.
// ------------------------------------------------------------------------------------
struct KmerSlot {           // the full description of a k-mer, 32 bytes: kept only for the few that need it (EXT)
  uint64_t key;             // KEY_INVALID: empty
  uint32_t on_a, on_b;      // on_cnt & KT_INLINE: (node rank, offset) of the only occurrence; else on_a = first entry of on_pos
  uint32_t off_a, off_b;    // off_cnt & KT_INLINE: (node rank, offset) of the only locus; else off_a = first locus entry,
                            // off_b = how many of the run's loci are at none of the on-path positions (those come first)
  uint32_t on_cnt, off_cnt; // occurrences on the indexed paths / starting loci with a k-walk spelling the k-mer
};
static_assert(sizeof(KmerSlot) == 32, "k-mer table slot must be 32 bytes");
constexpr uint32_t KT_INLINE = 0x80000000u;
constexpr uint32_t KT_OFFDUP = 0x40000000u;   // with KT_INLINE in off_cnt: the only locus is at one of the on-path positions

// What the query probes is a table of 16-BYTE slots: a divergent 16-byte load is the unit the
// memory pipeline charges for (two loads per probe cost twice: 27.9 G against 41.4 G probes/s,
// tools/rand_sector2.hip), and nearly every k-mer fits one: its single occurrence, its single
// locus, or both at the same position.  The rest keep a 32-byte record in a side array and pay a
// second access.
struct Slot16 {
  uint64_t kt;              // bits 0..61 the k-mer, bits 62..63 what (a, b) is
  uint32_t a, b;            // K16_ON1 / K16_OFF1 / K16_BOTH1: (node rank, offset); K16_EXT: a = index of the record
};
constexpr uint64_t K16_KEY = (1ull << 62) - 1;
constexpr uint64_t K16_ON1 = 0, K16_OFF1 = 1, K16_BOTH1 = 2, K16_EXT = 3;   // empty slot: (a, b) == (NIL, NIL)
constexpr uint32_t RES_INLINE = 0x80000000u, RES_EXT = 0x40000000u, RES_CNT = 0x3FFFFFFFu;

// any number of slots (the whole-genome table has to fit): slot = hash * n_slots / 2^64
struct KmerTableView { const Slot16* ht; uint64_t n_slots; const KmerSlot* ext; };
__device__ __forceinline__ uint64_t kt_home(uint64_t key, uint64_t n_slots) { return __umul64hi(mix64(key), n_slots); }
// Probe sequence: the four slots of the home slot's 64-byte sector first (cyclically, from the home slot),
// then the next sector's in the same order, and so on -- a second or third look costs no second sector
// (n_slots is a multiple of 4; t counts the looks so far).
__device__ __forceinline__ uint64_t kt_next(uint64_t h, uint32_t& t, uint64_t n_slots)
{
  ++t;
  const uint64_t in_sector = (h + 1) & 3ull;
  if (t & 3u) return (h & ~3ull) | in_sector;
  uint64_t b = (h & ~3ull) + 4;
  if (b >= n_slots) b = 0;
  return b | in_sector;
}

__device__ __forceinline__ uint64_t slot16_type(const KmerSlot& r)
{
  const bool on1 = r.on_cnt == (1u | KT_INLINE), off1 = (r.off_cnt & ~KT_OFFDUP) == (1u | KT_INLINE);
  if (on1 && r.off_cnt == 0) return K16_ON1;
  if (off1 && r.on_cnt == 0) return K16_OFF1;
  if (on1 && off1 && r.on_a == r.off_a && r.on_b == r.off_b) return K16_BOTH1;
  return K16_EXT;
}

// length of the run of equal keys starting at i (gallop, then bisect)
__device__ __forceinline__ uint64_t run_end(const uint64_t* __restrict__ keys, uint64_t n, uint64_t i)
{
  const uint64_t key = keys[i];
  uint64_t lo = i, stepw = 1;
  while (lo + stepw < n && keys[lo + stepw] == key) { lo += stepw; stepw <<= 1; }
  uint64_t hi = lo + stepw < n ? lo + stepw : n;
  while (hi - lo > 1) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  return hi;
}

// ---- construction of the 16-byte slots -----------------------------------------------------------------
// A table of full 32-byte records in between would take more than the device has at whole-genome size
// (6.4 G k-mers), so the slots are made straight from the two sorted k-mer streams: the path k-mers in
// suffix-array order and the sorted (k-mer, locus) pairs.  pk[row] = ((k-mer + 1) << 1) | 1 at rows whose
// suffix starts with a k-mer, carried forward with the low bit cleared at the others (a max-scan: valid
// k-mers are non-decreasing along the suffix array), so pk >> 1 is monotone and can be bisected.
__global__ void k_pk_encode(const uint32_t* __restrict__ sa, uint64_t n, uint32_t k, const uint64_t* __restrict__ text4,
                            uint64_t* __restrict__ pk)
{
  uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  uint32_t pos = sa[row];
  uint64_t key = 0;
  bool ok = (uint64_t)pos + k <= n;
  for (uint32_t i = 0; ok && i < k; ++i) {
    uint32_t a = pos + i;
    uint64_t nib = (text4[a >> 4] >> (60 - 4 * (a & 15))) & 0xFull;
    if (nib & 4) ok = false;
    key = (key << 2) | (nib & 3);
  }
  pk[row] = ok ? (((key + 1) << 1) | 1ull) : 0ull;
}

// after the max-scan: rows that are not the start of a k-mer keep the carried value with the low bit cleared
__global__ void k_pk_fix(const uint32_t* __restrict__ sa, uint64_t n, uint32_t k, const uint64_t* __restrict__ text4,
                         uint64_t* __restrict__ pk)
{
  uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  uint32_t pos = sa[row];
  bool ok = (uint64_t)pos + k <= n;
  for (uint32_t i = 0; ok && i < k; ++i) {
    uint32_t a = pos + i;
    if ((text4[a >> 4] >> (60 - 4 * (a & 15))) & 4ull) ok = false;
  }
  if (!ok) pk[row] &= ~1ull;
}

// first index in [lo, hi) with (a[i] >> sh) >= v
__device__ __forceinline__ uint64_t lower_bound_sh(const uint64_t* __restrict__ a, uint64_t lo, uint64_t hi, uint64_t v, uint32_t sh)
{
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((a[mid] >> sh) < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ void kt_place(Slot16* __restrict__ ht, uint64_t n_slots, uint64_t key, uint64_t type, uint32_t a, uint32_t b)
{
  const unsigned long long payload = (unsigned long long)a | ((unsigned long long)b << 32);
  uint64_t h = kt_home(key, n_slots);
  uint32_t t = 0;
  while (true) {
    unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&ht[h].a), ~0ull, payload);
    if (prev == ~0ull) { ht[h].kt = key | (type << 62); return; }
    h = kt_next(h, t, n_slots);
  }
}

// The path k-mer streams of the index's parts (one part unless the text passes the 32-bit row limit).
struct PkPart {
  const uint64_t* pk; uint64_t n;                    // encoded k-mers along the part's suffix array
  const uint32_t* sa; const SegRec* seg; const uint32_t* seg_rank; const uint32_t* seg_dir;
};
struct PkParts { PkPart p[PSIGPU_MAX_PARTS]; uint32_t n_parts; };
constexpr uint32_t KT_DEDUP_MAX = 32;      // runs of up to this many occurrences are de-duplicated at build time

// rows [first, first + count) of `key` in a part (count 0: not a k-mer of this part)
__device__ __forceinline__ uint64_t pk_run(const PkPart& pt, uint64_t key, uint64_t* first)
{
  const uint64_t at = lower_bound_sh(pt.pk, 0, pt.n, key + 1, 1);
  if (at >= pt.n || (pt.pk[at] >> 1) != key + 1) return 0;
  const uint64_t r_end = lower_bound_sh(pt.pk, at + 1, pt.n, key + 2, 1);
  uint64_t lo = at + 1, hi = r_end;                  // first row in (at, r_end) with the low bit clear (a carried value)
  while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if (pt.pk[mid] & 1ull) lo = mid + 1; else hi = mid; }
  *first = at;
  return lo - at;
}

__device__ __forceinline__ uint2 pk_position(const PkPart& pt, uint64_t row)
{
  const uint32_t p = pt.sa[row];
  uint32_t d = pt.seg_dir[p >> DIR_SHIFT];
  while (pt.seg[d + 1].start <= p) ++d;
  return make_uint2(pt.seg_rank[d], pt.seg[d].noff + (p - pt.seg[d].start));
}

// One thread per suffix-array row of part `q`; the first row of every run of equal path k-mers makes the
// k-mer's slot -- unless an earlier part holds the k-mer too (that part makes it) -- with the k-mer's
// occurrences in the later parts and what the starting loci contribute to it (bisection in the sorted
// pairs).  A k-mer with one occurrence keeps its position in the slot; the positions of the others go to
// `on_pos` (a run per k-mer), so that a query needs nothing of the FM parts.
// FILL = false: only count the k-mers that need a 32-byte record (EXT), the positions, the path k-mers.
template <bool FILL>
__global__ void k_kt_direct_on(PkParts parts, uint32_t q, const uint64_t* __restrict__ okeys,
                               uint32_t* ovals, uint64_t n_off, const uint2* __restrict__ loci,
                               Slot16* __restrict__ ht, uint64_t n_slots, KmerSlot* __restrict__ ext,
                               uint2* __restrict__ on_pos, unsigned long long* __restrict__ cnt /* [0] EXT records, [1] path k-mers, [2] positions */,
                               bool dedup)
{
  const PkPart& me = parts.p[q];
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= me.n) return;
  const uint64_t v = me.pk[i];
  if (!(v & 1ull) || (i != 0 && (me.pk[i - 1] >> 1) == (v >> 1))) return;       // not the first row of a k-mer
  const uint64_t key = (v >> 1) - 1;
  uint64_t first[PSIGPU_MAX_PARTS], count[PSIGPU_MAX_PARTS], on_cnt = 0;
  for (uint32_t r = 0; r < q; ++r)
    if (pk_run(parts.p[r], key, &first[r])) return;                              // an earlier part owns this k-mer
  for (uint32_t r = q; r < parts.n_parts; ++r) { count[r] = pk_run(parts.p[r], key, &first[r]); on_cnt += count[r]; }
  if (!FILL) atomicAdd(&cnt[1], 1ull);
  const uint64_t j = lower_bound_sh(okeys, 0, n_off, key, 0);
  uint64_t off_cnt = 0;
  if (j < n_off && okeys[j] == key) off_cnt = lower_bound_sh(okeys, j + 1, n_off, key + 1, 0) - j;
  KmerSlot r;
  r.key = key; r.on_a = r.on_b = r.off_a = r.off_b = 0;
  uint2 u[KT_DEDUP_MAX];               // the k-mer's on-path positions, when they are few
  uint32_t n_u = 0;
  bool have_u = false;
  if (on_cnt == 1) {
    const uint2 at = pk_position(me, i);
    r.on_a = at.x; r.on_b = at.y; r.on_cnt = 1u | KT_INLINE;
    u[0] = at; n_u = 1; have_u = true;
  } else if (dedup && on_cnt <= KT_DEDUP_MAX) {
    // The same graph position on several indexed paths (full paths: nearly every k-mer) is one hit: kept
    // once, and a k-mer whose occurrences are all one position stays in its slot like a single occurrence.
    // (Not with a gocc threshold, which counts occurrences in the path text: index_iter.hpp:843-847.)
    have_u = true;
    for (uint32_t pr = q; pr < parts.n_parts; ++pr)
      for (uint64_t t = 0; t < count[pr]; ++t) {
        const uint2 at = pk_position(parts.p[pr], first[pr] + t);
        bool seen = false;
        for (uint32_t x = 0; x < n_u; ++x) seen = seen || (u[x].x == at.x && u[x].y == at.y);
        if (!seen) u[n_u++] = at;
      }
    if (n_u == 1) { r.on_a = u[0].x; r.on_b = u[0].y; r.on_cnt = 1u | KT_INLINE; }
    else {
      const unsigned long long base = atomicAdd(&cnt[2], (unsigned long long)n_u);
      if (FILL) for (uint32_t x = 0; x < n_u; ++x) on_pos[base + x] = u[x];
      r.on_a = (uint32_t)base; r.on_cnt = n_u;
    }
  } else {
    const unsigned long long base = atomicAdd(&cnt[2], (unsigned long long)on_cnt);
    if (FILL) {
      uint64_t w = base;
      for (uint32_t pr = q; pr < parts.n_parts; ++pr)
        for (uint64_t t = 0; t < count[pr]; ++t) on_pos[w++] = pk_position(parts.p[pr], first[pr] + t);
    }
    r.on_a = (uint32_t)base; r.on_cnt = (uint32_t)min(on_cnt, (uint64_t)0x3FFFFFFFu);
  }
  // A locus at one of the on-path positions gives the hit the path gives (the on-path k-walk from an
  // uncovered locus): when the on-path occurrences are emitted it is left out -- loci of that kind go to
  // the end of the k-mer's run, and the record says how many are in front of them.
  auto on_path_position = [&](uint2 lc) {
    bool seen = false;
    for (uint32_t x = 0; x < n_u; ++x) seen = seen || (u[x].x == lc.x && u[x].y == lc.y);
    return seen;
  };
  if (off_cnt == 1) {
    const uint2 lc = loci[ovals[j]];
    r.off_a = lc.x; r.off_b = lc.y; r.off_cnt = 1u | KT_INLINE;
    if (have_u && on_path_position(lc)) r.off_cnt |= KT_OFFDUP;
  } else {
    uint64_t front = off_cnt;
    if (have_u && off_cnt <= KT_DEDUP_MAX) {
      front = 0;
      for (uint64_t t = 0; t < off_cnt; ++t) {
        const uint32_t v = ovals[j + t];
        if (on_path_position(loci[v])) continue;
        if (FILL && front != t) { ovals[j + t] = ovals[j + front]; ovals[j + front] = v; }
        ++front;
      }
    }
    r.off_a = (uint32_t)j; r.off_b = (uint32_t)front; r.off_cnt = (uint32_t)off_cnt;
  }
  const uint64_t type = slot16_type(r);
  if (type == K16_EXT) {
    const unsigned long long e = atomicAdd(&cnt[0], 1ull);
    if (FILL) { ext[e] = r; kt_place(ht, n_slots, key, type, (uint32_t)e, 0); }
  } else if (FILL) kt_place(ht, n_slots, key, type, r.on_a, r.on_b);
}

// One thread per sorted (k-mer, locus) pair; the first pair of every run whose k-mer is NOT a path k-mer
// makes the slot (the others were made by k_kt_direct_on).
template <bool FILL>
__global__ void k_kt_direct_off(const uint64_t* __restrict__ okeys, const uint32_t* __restrict__ ovals, uint64_t n_off,
                                const uint2* __restrict__ loci, PkParts parts,
                                Slot16* __restrict__ ht, uint64_t n_slots, KmerSlot* __restrict__ ext,
                                unsigned long long* __restrict__ cnt)
{
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_off) return;
  const uint64_t key = okeys[j];
  if (j && okeys[j - 1] == key) return;
  for (uint32_t r = 0; r < parts.n_parts; ++r) {
    uint64_t f;
    if (pk_run(parts.p[r], key, &f)) return;             // a path k-mer: its slot holds the loci too
  }
  const uint64_t off_cnt = lower_bound_sh(okeys, j + 1, n_off, key + 1, 0) - j;
  if (off_cnt == 1) {
    if (FILL) { const uint2 lc = loci[ovals[j]]; kt_place(ht, n_slots, key, K16_OFF1, lc.x, lc.y); }
  } else {
    const unsigned long long e = atomicAdd(&cnt[0], 1ull);
    if (FILL) {
      KmerSlot r;
      r.key = key; r.on_a = r.on_b = 0; r.on_cnt = 0; r.off_a = (uint32_t)j; r.off_b = (uint32_t)off_cnt; r.off_cnt = (uint32_t)off_cnt;
      ext[e] = r;
      kt_place(ht, n_slots, key, K16_EXT, (uint32_t)e, 0);
    }
  }
}

// The whole of K1 in this mode: one probe per seed -- one 16-byte load -- over the wave ranges of
// K2.  Leaves 16 bytes per seed for K2: (a, b) of the slot, the number of on-path occurrences and
// of loci that are wanted (phase flags, gocc threshold), and whether they are the inline position.
//
// R8 (round 4): 8 bytes of results per seed instead of 16 -- [ext flag | locus flag | on-path flag | offset (28 bits) | node rank
// or record number (32 bits)]: the probe shares the load path with its own result stream, and nearly every seed is answered
// from its slot (one position, inline).  A seed answered by a 32-byte record hands on the record's number only; the emit
// kernel reads the record anyway and applies the phase flags and the threshold again.  Graphs with a node of 2^28 bases or
// more keep the 16-byte form (option "res16": always).
constexpr uint64_t R8_ON = 1ull << 60, R8_OFF = 1ull << 61, R8_EXT = 1ull << 62;
constexpr uint32_t R8_NOFF_BITS = 28;

// counts of a k-mer's 32-byte record under the call's phase flags and threshold: (on-path occurrences emitted, loci emitted)
__device__ __forceinline__ uint2 ext_counts(const uint4 e /* off_a, off_b, on_cnt, off_cnt */, bool want_on, bool want_off, uint32_t gocc_thr)
{
  const uint32_t c_on = e.z & ~KT_INLINE;
  const bool on_emitted = want_on && c_on <= gocc_thr;
  uint2 r = make_uint2(on_emitted ? min(c_on, RES_CNT) : 0u, 0u);
  // loci at an on-path position are left out when the on-path occurrences are emitted
  if (want_off) r.y = (e.w & KT_INLINE) ? ((on_emitted && (e.w & KT_OFFDUP)) ? 0u : 1u) : (on_emitted ? e.y : e.w);
  return r;
}

// One seed's look-up: the slot of `key` found along the probe sequence that starts at `h` with the slot `v` already
// loaded (the caller issues the first load of several seeds before it looks at any: their latencies overlap).  Returns what
// k_kmer_emit turns into records: (x, y) the position or the record's index, z / w the on-path and off-path counts.
__device__ __forceinline__ uint4 kt_resolve(const KmerTableView& kt, uint64_t key, uint64_t h, uint4 v, bool want_on, bool want_off,
                                            uint32_t gocc_thr)
{
  uint4 res = make_uint4(0, 0, 0, 0);
  uint32_t t = 0;
  while (true) {
    const uint64_t w = (uint64_t)v.x | ((uint64_t)v.y << 32);
    const bool empty = v.z == NIL && v.w == NIL;          // (an all-T 31-mer with an EXT record is all ones in w)
    if (!empty && (w & K16_KEY) == key) {
      const uint64_t type = w >> 62;
      res.x = v.z; res.y = v.w;
      if (type == K16_EXT) {
        const uint4 e = load16(reinterpret_cast<const uint4*>(kt.ext + v.z) + 1);      // off_a, off_b, on_cnt, off_cnt
        const uint2 cc = ext_counts(e, want_on, want_off, gocc_thr);
        res.z = RES_EXT | cc.x;
        res.w = cc.y;
      } else {
        if (want_on && type != K16_OFF1) res.z = 1u | RES_INLINE;
        // (one occurrence and one locus at the same position, both phases asked for: one hit)
        if (want_off && type != K16_ON1 && !(want_on && type == K16_BOTH1)) res.w = 1u | RES_INLINE;
      }
      break;
    }
    if (empty) break;
    h = kt_next(h, t, kt.n_slots);
    v = load16(kt.ht + h);
  }
  return res;
}

template <bool R8>
__global__ void __launch_bounds__(256)
k_kmer_probe(KmerTableView kt, const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params,
             uint64_t seeds_cap, uint32_t per_wave, bool want_on, bool want_off, uint32_t gocc_thr,
             uint4* __restrict__ seed_res, uint64_t* __restrict__ wave_total, uint64_t* __restrict__ wave_total_off,
             DevCounters* ctr)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  uint64_t wsum = 0, osum = 0;
  uint32_t n_live = 0;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t seed = base + lane;
    if (seed >= s1) continue;
    const uint64_t key = seed_key[seed];
    uint4 res = make_uint4(0, 0, 0, 0);
    if (key != KEY_INVALID) {
      const uint64_t h = kt_home(key, kt.n_slots);
      res = kt_resolve(kt, key, h, load16(kt.ht + h), want_on, want_off, gocc_thr);
    }
    if constexpr (R8) {
      uint64_t r8 = (uint64_t)res.x;
      if (res.z & RES_EXT) r8 |= R8_EXT;
      else r8 |= ((uint64_t)res.y << 32) | ((res.z & RES_CNT) ? R8_ON : 0ull) | ((res.w & ~RES_INLINE) ? R8_OFF : 0ull);
      reinterpret_cast<uint64_t*>(seed_res)[seed] = r8;
    } else
    seed_res[seed] = res;
    const uint32_t con = res.z & RES_CNT, coff = res.w & ~RES_INLINE;
    wsum += con; osum += coff;
    n_live += con != 0;
  }
  for (int d = 32; d > 0; d >>= 1) {
    wsum += __shfl_down(wsum, d); osum += __shfl_down(osum, d); n_live += __shfl_down(n_live, d);
  }
  if (lane == 0) {
    wave_total[wave] = wsum;
    wave_total_off[wave] = osum;
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
  }
}

// graph position of every suffix-array row (whole SA resident): locate in one access for the
// occurrences that are not covered by SaRec / an inline slot (repeats, several indexed paths)
__global__ void k_build_saloc(const uint32_t* __restrict__ sa, uint64_t n, const SegRec* __restrict__ seg,
                              const uint32_t* __restrict__ seg_rank, const uint32_t* __restrict__ seg_dir,
                              uint2* __restrict__ out)
{
  uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint32_t p = sa[row];
  uint32_t d = seg_dir[p >> DIR_SHIFT];
  while (seg[d + 1].start <= p) ++d;
  out[row] = make_uint2(seg_rank[d], seg[d].noff + (p - seg[d].start));
}

// per-row records (SaRec) for seed length k = q + rem
__global__ void k_build_sarec(const uint32_t* __restrict__ sa, uint64_t n, uint32_t rem, const SegRec* __restrict__ seg,
                              const uint32_t* __restrict__ seg_rank, const uint32_t* __restrict__ seg_dir,
                              const uint64_t* __restrict__ text4, SaRec* __restrict__ out)
{
  uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint32_t pos = sa[row];
  uint64_t bits = 0, valid = 0;
  for (uint32_t i = 1; i <= 29 && i <= pos; ++i) {
    uint32_t a = pos - i;
    uint64_t nib = (text4[a >> 4] >> (60 - 4 * (a & 15))) & 0xFull;
    if (nib & 4) break;
    bits |= (nib & 3) << (2 * (i - 1));
    ++valid;
  }
  SaRec r = { 0, 0, bits | (valid << 58) };
  if (valid >= rem) {
    uint32_t p = pos - rem;
    uint32_t d = seg_dir[p >> DIR_SHIFT];
    while (seg[d + 1].start <= p) ++d;
    r.node = seg_rank[d];
    r.noff = seg[d].noff + (p - seg[d].start);
  }
  out[row] = r;
}

// The interval table with the first row's record inside its entries: one 32-byte entry per q-mer -- (l, r) and
// the SaRec of row l -- so that K1 of the FM modes learns a seed's interval AND verifies its first row (the only
// one for most q-mers: 1.45 rows on average) from ONE sector.  k_fm_search_direct runs at the fabric's request
// rate; this takes one of its ~3.4 requests per seed away.
struct FtabX { uint32_t l, r, node, noff; uint64_t ctx, pad; };
static_assert(sizeof(FtabX) == 32, "two 16-byte loads from one sector");

__global__ void k_build_ftabx(const uint2* __restrict__ ftab, uint64_t n_entries, const SaRec* __restrict__ sarec, FtabX* __restrict__ out)
{
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_entries) return;
  const uint2 iv = ftab[c];
  FtabX e = { iv.x, iv.y, 0, 0, 0, 0 };
  if (iv.y > iv.x) { const SaRec rec = sarec[iv.x]; e.node = rec.node; e.noff = rec.noff; e.ctx = rec.ctx; }
  uint4* o = reinterpret_cast<uint4*>(out + c);
  o[0] = make_uint4(e.l, e.r, e.node, e.noff);
  o[1] = make_uint4((uint32_t)e.ctx, (uint32_t)(e.ctx >> 32), 0, 0);
}

// 16 text symbols (4 bits each, first on top) starting `rem` symbols in front of `pos`; both words
// are always loaded (the text carries two words of padding), so several windows can be in flight
__device__ __forceinline__ bool window_matches(uint64_t w0, uint64_t w1, uint32_t a, uint32_t rem, uint64_t key, uint32_t k)
{
  uint32_t sh = (a & 15) * 4;
  uint64_t x = w0 << sh;
  if (sh) x |= w1 >> (64 - sh);
  uint64_t top = rem == 16 ? ~0ull : ~(~0ull >> (4 * rem));
  if (x & top & 0x4444444444444444ull) return false;    // a separator / the sentinel
  uint64_t y = x & 0x3333333333333333ull;               // nibbles -> 2-bit codes, order kept
  y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  y = (y | (y >> 4)) & 0x00FF00FF00FF00FFull;
  y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
  y = (y | (y >> 16)) & 0x00000000FFFFFFFFull;
  uint32_t got = (uint32_t)y >> (32 - 2 * rem);
  uint32_t want = (uint32_t)(key >> (2 * (k - rem)));
  return got == want;
}

// K1 when the interval table and the per-row records (SaRec) are resident: no LF step is needed
// for a seed whose q-mer interval has at most VERIFY_ROWS rows -- look the interval up, compare
// the bases in front of each row (they are in the row's record) with the head of the seed.  There
// is nothing for a quad to share, so this is one lane per seed, 64 seeds per wave round, and the
// lane's independent loads are issued together: the interval-table entry, the locus k-mer table
// slot (the probe that answers seeds_off_paths, when that table is in use), then the rows'
// records four at a time.  Seeds with a larger interval are appended to `defer` for k_fm_search
// (quad kernel, list mode).
__global__ void __launch_bounds__(256)
k_fm_search_direct(FMView fm, const FtabX* __restrict__ ftabx, LktView lk, const uint64_t* __restrict__ seed_key, const uint64_t* __restrict__ params,
                   uint64_t seeds_cap, uint32_t per_wave, uint32_t k, uint32_t gocc_thr, SeedOut so,
                   uint64_t* __restrict__ wave_total, uint64_t* __restrict__ wave_total_off,
                   uint32_t* __restrict__ defer, DevCounters* ctr)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  const uint32_t q = fm.ftab_len, rem = k - q;
  const uint64_t qmask = (1ull << (2 * q)) - 1ull;
  const uint64_t wmask = rem ? (1ull << (2 * rem)) - 1ull : 0ull;
  uint32_t n_live = 0, n_rows = 0;
  uint64_t wsum = 0, osum = 0;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t seed = base + lane;
    const bool in = seed < s1;
    const uint64_t key = in ? seed_key[seed] : KEY_INVALID;
    const bool valid = key != KEY_INVALID;
    TableSlot sl = { KEY_INVALID, 0, 0 };
    uint64_t h = 0;
    const bool probing = lk.ht != nullptr && valid;
    if (probing) { h = lkt_home(key, lk.n_slots); sl = lk.ht[h]; }
    uint32_t l = 0, r = 0;
    SaRec first = { 0, 0, 0 };                  // row l's record, when the interval table carries it
    if (valid) {
      if (ftabx) {
        const uint4* e = reinterpret_cast<const uint4*>(ftabx + (key & qmask));
        const uint4 a = e[0], b = e[1];           // (one sector)
        l = a.x; r = a.y; first.node = a.z; first.noff = a.w; first.ctx = (uint64_t)b.x | ((uint64_t)b.y << 32);
      } else { uint2 iv = fm.ftab[key & qmask]; l = iv.x; r = iv.y; }
    }
    uint32_t cnt = r > l ? r - l : 0u, aux = 0, on_node = 0, on_noff = 0;
    const bool deferred = rem != 0 && cnt > VERIFY_ROWS;
    if (rem != 0 && cnt != 0 && !deferred) {
      // the rem bases in front of each row against the head of the seed, rows four at a time; the
      // first matching row's record also gives K2 the hit itself
      const uint64_t want = key >> (2 * q);
      uint32_t mask = 0;
      for (uint32_t t0 = 0; t0 < cnt; t0 += 4) {
        SaRec c[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          c[j] = SaRec{ 0, 0, 0 };
          if (t0 + j < cnt) {
            if (ftabx && t0 + j == 0) { c[j] = first; continue; }
            uint4 v = *reinterpret_cast<const uint4*>(&fm.sarec[l + t0 + j]);
            c[j].node = v.x; c[j].noff = v.y; c[j].ctx = (uint64_t)v.z | ((uint64_t)v.w << 32);
          }
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j)
          if (t0 + j < cnt && (c[j].ctx >> 58) >= rem && (c[j].ctx & wmask) == want) {
            if (mask == 0) { on_node = c[j].node; on_noff = c[j].noff; aux = AUX_RESOLVED; }
            mask |= 1u << (t0 + j);
          }
      }
      n_rows += cnt;
      cnt = (uint32_t)__popc(mask);
      aux |= (rem << 8) | mask;
    }
    // the locus k-mer table: the first probe is back by now; collisions are rare
    uint32_t ofirst = 0, ocnt = 0, onoff = 0;
    if (probing) lkt_resolve(lk, key, h, sl, ofirst, ocnt, onoff);
    const bool keep = !deferred && cnt != 0 && cnt <= gocc_thr;
    if (in) {
      so.iv_lo[seed] = l;
      so.iv_cnt[seed] = keep ? cnt : 0u;
      so.iv_aux[seed] = aux;
      so.on_node[seed] = on_node;
      so.on_noff[seed] = on_noff;
      if (lk.ht != nullptr) { so.off_first[seed] = ofirst; so.off_cnt[seed] = ocnt; so.off_noff[seed] = onoff; }
      n_live += keep;
      wsum += keep ? cnt : 0u;
      osum += ocnt & ~OFF_INLINE;
    }
    uint64_t dm = __ballot(deferred);
    if (dm) {
      unsigned long long at = 0;
      if (lane == 0) at = atomicAdd(&ctr->n_defer.v, (unsigned long long)__popcll(dm));
      at = __shfl(at, 0);
      if (deferred) defer[at + __popcll(dm & lanemask_lt())] = (uint32_t)seed;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_live += __shfl_down(n_live, d); wsum += __shfl_down(wsum, d);
    osum += __shfl_down(osum, d); n_rows += __shfl_down(n_rows, d);
  }
  if (lane == 0) {
    wave_total[wave] = wsum;
    if (wave_total_off) wave_total_off[wave] = osum;
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
    if (n_rows) ctr->n_rows_verified.add((unsigned long long)n_rows);
  }
}

// ------------------------------------------------------------------------------------
// K2 for a sampled suffix array (sa_rate > 1), in two kernels.
//
// k_fm_walk: every on-path occurrence is LF-walked to a sampled row (csa[i] behind fmindex.hpp:734-748).
// With SA-order sampling -- rows i % s == 0 keep their value, as sdsl's csa_wt<wt_huff<>, 32, 64> does -- a
// walk ends with probability 1/s per step: lengths are geometric, mean s - 1, and the longest of 16 is
// about 3.4 times the mean.  A kernel that keeps the 16 quads of a wave in step (rounds 1-2) runs at a fifth
// of the rate the walks themselves allow (9.5 ms against 3.0 for the pair below, profiles/r03_lf_ab_locate.jsonl);
// here the quads are decoupled.  A wave stages 64 seeds of its range in
// registers (interval, count, first output slot: one coalesced load and one wave scan per 64 seeds), and a
// quad whose walk has ended takes the next staged seed through shuffles -- no memory access on that path --
// so every quad issues exactly one sector request per iteration (a rank block, or the sample that ends the
// walk) whatever the others are doing.  Out: 12 bytes per hit (text position | occurrence number, seed).
//
// k_hits_resolve: one lane per hit -- text position -> segment -> (node, offset), or the locus of a table
// hit -- and the 32-byte record (StringSet::get_position sequence.hpp:539-546 + position_to_id/offset
// pathindex.hpp:378-416 in one step).  Independent lanes, nothing to wait for but their own loads.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_fm_walk(FMView fm, const uint32_t* __restrict__ samples, uint32_t sa_rate, const uint32_t* __restrict__ exc_sa,
          const uint32_t* __restrict__ iv_lo, const uint32_t* __restrict__ iv_cnt, const uint32_t* __restrict__ off_cnt,
          const uint64_t* __restrict__ wave_off, const uint64_t* __restrict__ params, uint64_t seeds_cap, uint32_t per_wave,
          uint64_t* __restrict__ hit_a, uint32_t* __restrict__ hit_seed, uint64_t cap, DevCounters* ctr)
{
  __shared__ uint8_t sel_all[4][64];             // per wave: the staged seeds that have on-path occurrences, compacted
  __shared__ uint32_t s_sup[SUP_LDS];
  stage_exc_super(fm, s_sup);
  const uint32_t lane = lane_id(), ql = lane & 3, wib = threadIdx.x >> 6;
  uint8_t* sel = sel_all[wib];
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_items = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_items, s0 + per_wave);
  uint64_t woff = s0 < s1 ? wave_off[wave] : 0;       // next output slot of this wave
  uint64_t cursor = s0, win_base = s0;                // seeds [win_base, win_base + 64) are staged; cursor = next to stage
  uint32_t w_lo = 0, w_cnt = 0, win_n = 0, taken = 0;
  uint64_t w_out0 = 0;
  bool have = false;
  uint32_t row = 0, steps = 0, occ = 0, q_lo = 0, q_cnt = 0, q_seed = 0, n_walk = 0;
  uint64_t q_out0 = 0;
  const uint64_t leaders = 0x1111111111111111ull;      // lane 0 of every quad
  while (true) {
    const uint64_t nm = __ballot(!have) & leaders;       // quads without a walk
    if (nm) {
      if (taken == win_n && cursor < s1) {
        // stage the next 64 seeds; the table hits among them are described right here (no walk)
        const uint64_t item = cursor + lane;
        const bool in = item < s1;
        w_lo = in ? iv_lo[item] : 0u;
        w_cnt = in ? iv_cnt[item] : 0u;
        const uint32_t coff = (in && off_cnt) ? (off_cnt[item] & ~OFF_INLINE) : 0u;
        uint32_t incl = w_cnt + coff;
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = (uint32_t)__shfl_up((int)incl, d);
          if (lane >= (uint32_t)d) incl += t;
        }
        w_out0 = woff + (incl - (w_cnt + coff));
        woff += (uint32_t)__shfl((int)incl, 63);
        for (uint32_t o = 0; o < coff; ++o) {
          const uint64_t h = w_out0 + w_cnt + o;
          if (h < cap) { hit_a[h] = (uint64_t)(w_cnt + o) << 32; hit_seed[h] = (uint32_t)item; }
        }
        const uint64_t m = __ballot(w_cnt != 0);
        if (w_cnt) sel[__popcll(m & lanemask_lt())] = (uint8_t)lane;
        win_n = (uint32_t)__popcll(m); taken = 0;
        win_base = cursor;
        cursor += 64;
        __builtin_amdgcn_wave_barrier();
      }
      // quads without a walk take staged seeds in order
      const uint32_t idx = taken + (uint32_t)__popcll(nm & ((1ull << (lane & ~3u)) - 1ull));     // quads in front that also take one
      const bool gets = !have && idx < win_n;
      const int src = gets ? (int)sel[idx] : 0;
      const uint32_t lo_ = (uint32_t)__shfl((int)w_lo, src), cnt_ = (uint32_t)__shfl((int)w_cnt, src);
      const uint64_t out_ = __shfl(w_out0, src);
      if (gets) { have = true; q_lo = lo_; q_cnt = cnt_; q_out0 = out_; q_seed = (uint32_t)(win_base + (uint32_t)src); occ = 0; row = lo_; steps = 0; }
      taken = min(win_n, taken + (uint32_t)__popcll(nm));
    }
    if (!__any(have)) {
      if (cursor >= s1 && taken == win_n) break;
      continue;
    }
    // ---- one sector request per walking quad: the rank block of its row, or the sample that ends the walk ----
    if (have) {
      bool done = false;
      uint32_t pos = 0;
      if ((row & (sa_rate - 1)) == 0) {
        pos = samples[row / sa_rate] + steps;
        done = true;
      } else {
        const uint32_t blk = row / BLOCK_SYMS, off = row - blk * BLOCK_SYMS;
        const uint4 v = fm.blocks[(uint64_t)blk * 4 + ql];
        uint32_t sym = 0;                         // BWT[row]: the owning lane extracts it, the quad sum hands it round
        if (ql == 1 + off / 64) {
          const uint32_t o = off & 63;
          const uint32_t lo = o < 32 ? v.x : v.y, hi = o < 32 ? v.z : v.w;
          sym = ((lo >> (o & 31)) & 1u) | (((hi >> (o & 31)) & 1u) << 1);
        }
        sym = quad_sum(sym);
        uint32_t ex = 0;                          // a separator / the sentinel in the BWT: its SA value is stored
        if (ql == 0 && (v.w & 0xFF) != 0) {
          const uint32_t e0 = (v.w >> 8) + exc_super(fm, s_sup, blk), ne = v.w & 0xFF;
          const uint32_t end = (ne == 255) ? fm.n_exc : e0 + ne;
          for (uint32_t q = e0; q < end; ++q) {
            const uint32_t rr = fm.exc_row[q];
            if (rr == row) { ex = q + 1; break; }
            if (rr > row) break;
          }
        }
        ex = quad_bcast0(ex);
        if (ex) { pos = exc_sa[ex - 1] + steps; done = true; }
        else { row = fm.C[sym] + quad_rank(fm, s_sup, v, ql, sym, row); ++steps; n_walk += ql == 0; }
      }
      if (done) {
        const uint64_t h = q_out0 + occ;
        if (ql == 0 && h < cap) { hit_a[h] = (uint64_t)pos | ((uint64_t)occ << 32); hit_seed[h] = q_seed; }
        ++occ;
        if (occ < q_cnt) { row = q_lo + occ; steps = 0; }
        else have = false;
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) n_walk += __shfl_down(n_walk, d);
  if (lane == 0 && n_walk) ctr->n_locate_steps.add((unsigned long long)n_walk);
}

__global__ void __launch_bounds__(256)
k_hits_resolve(MapView mv, const uint64_t* __restrict__ hit_a, const uint32_t* __restrict__ hit_seed,
               const uint32_t* __restrict__ iv_cnt, const uint32_t* __restrict__ off_first, const uint32_t* __restrict__ off_cnt,
               const uint32_t* __restrict__ off_noff, const LocusEnt* __restrict__ ent, const uint64_t* __restrict__ range,
               const uint2* __restrict__ seed_info, uint64_t rec_offset, psigpu_hit* __restrict__ hits, uint64_t cap)
{
  // range[0], range[1]: the output slots of this part of the index (k_wave_offsets)
  const uint64_t n = min(range[1], cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t h = range[0] + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; h < n; h += stride) {
    const uint64_t a = hit_a[h];
    const uint32_t seed = hit_seed[h], occ = (uint32_t)(a >> 32), pos = (uint32_t)a;
    const uint32_t con = iv_cnt[seed];
    const uint2 si = seed_info[seed];
    uint64_t nid, noff;
    if (occ < con) {
      uint32_t d = mv.seg_dir[pos >> DIR_SHIFT];
      while (mv.seg[d + 1].start <= pos) ++d;
      const SegRec sr = mv.seg[d];
      nid = sr.node_id; noff = (uint64_t)sr.noff + (pos - sr.start);
    } else {
      uint2 lc = make_uint2(off_first[seed], 0);
      if (off_cnt[seed] & OFF_INLINE) lc.y = off_noff[seed];
      else lc = mv.loci[ent[lc.x + (occ - con)]];
      nid = mv.id_affine ? mv.id_base + lc.x : mv.node_id[lc.x]; noff = lc.y;
    }
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(hits + h);
    dst[0] = make_ulonglong2(nid, noff);
    dst[1] = make_ulonglong2(rec_offset + si.x, (uint64_t)si.y);
  }
}

// One hit of a seed: occurrence `occ` of its `con` on-path rows, or entry occ - con of its run in
// the locus k-mer table.
struct SeedHits {
  uint32_t lo, con, aux, on_node, on_noff, ofirst, ocnt, onoff;
};

__device__ __forceinline__ void resolve_hit(const MapView& mv, const LocusEnt* __restrict__ ent, const SeedHits& sh,
                                            uint32_t occ, uint64_t& nid, uint64_t& noff)
{
  if (occ < sh.con) {
    uint32_t row = sh.lo + occ, rows = sh.aux & 0xFFu, rem = (sh.aux >> 8) & 0xFFu;
    if (rows) {
      for (uint32_t i = 0; i < occ; ++i) rows &= rows - 1;
      row = sh.lo + (uint32_t)__ffs((int)rows) - 1;
    }
    if ((sh.aux & AUX_RESOLVED) && occ == 0) {
      nid = mv.id_affine ? mv.id_base + sh.on_node : mv.node_id[sh.on_node];
      noff = sh.on_noff;
    } else if (sh.aux & AUX_ONPOS) {
      const uint2 at = mv.on_pos[sh.lo + occ];
      nid = mv.id_affine ? mv.id_base + at.x : mv.node_id[at.x];
      noff = at.y;
    } else if (mv.sarec != nullptr && rem == mv.sarec_rem) {
      // verified by K1 against this row's record: it names the seed's first base
      uint2 at = *reinterpret_cast<const uint2*>(&mv.sarec[row]);
      nid = mv.id_affine ? mv.id_base + at.x : mv.node_id[at.x];
      noff = at.y;
    } else if (mv.saloc != nullptr && rem == 0) {
      uint2 at = mv.saloc[row];
      nid = mv.id_affine ? mv.id_base + at.x : mv.node_id[at.x];
      noff = at.y;
    } else {
      uint32_t pos = mv.samples[row] - rem;
      uint32_t d = mv.seg_dir[pos >> DIR_SHIFT];
      while (mv.seg[d + 1].start <= pos) ++d;
      SegRec sr = mv.seg[d];
      nid = sr.node_id; noff = (uint64_t)sr.noff + (pos - sr.start);
    }
  } else if (sh.ocnt & OFF_INLINE) {
    nid = mv.id_affine ? mv.id_base + sh.ofirst : mv.node_id[sh.ofirst];
    noff = sh.onoff;
  } else {
    const uint2 lc = mv.loci[ent[sh.ofirst + (occ - sh.con)]];
    nid = mv.id_affine ? mv.id_base + lc.x : mv.node_id[lc.x];
    noff = lc.y;
  }
}

// One round of emission, shared by K2 of the FM modes and of the k-mer table mode: 64 seeds, one per lane,
// each with `cnt` hits described by `sh`.  When no seed of the round has more than two hits (the usual
// case) every lane writes its own: the records of consecutive lanes are consecutive.  Otherwise the
// round's HITS are handed out to the lanes 64 at a time: lane j finds the seed that owns hit j by
// bisecting the prefix of the counts (shuffles), so a seed with many occurrences is spread over the wave
// instead of serialising one lane.  `woff` (wave-uniform) is the wave's next output slot.
__device__ __forceinline__ void emit_round(const MapView& mv, const LocusEnt* __restrict__ ent, const SeedHits& sh, uint32_t cnt,
                                           uint2 si, uint64_t& woff, uint64_t rec_offset, psigpu_hit* __restrict__ hits, uint64_t cap,
                                           bool transpose = true)
{
  const uint32_t lane = lane_id();
  if (transpose && __all(cnt == 1) && woff + 64 <= cap) {
    // One hit per seed (the usual round; asked before the prefix sums, which it does not need), the round's 64 records
    // are 2 KB in a row.  A lane storing its own record stores two 16-byte halves 32 bytes apart -- an instruction covers
    // half of every line it touches -- so the records are transposed by shuffles first: lane l stores half l & 1 of
    // record l >> 1 (then of record 32 + (l >> 1)): 1 KB per instruction without holes (tools/probe_shape.hip: 4.3 ->
    // 5.4 TB/s for this shape).  Offsets in nodes and reads are 32-bit values.
    uint64_t nid, noff;
    resolve_hit(mv, ent, sh, 0, nid, noff);
    const uint64_t rid = rec_offset + si.x;
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(hits + woff);
    const bool second = lane & 1u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int src = 32 * h + (int)(lane >> 1);
      const uint32_t a0 = (uint32_t)__shfl((int)(uint32_t)nid, src), a1 = (uint32_t)__shfl((int)(uint32_t)(nid >> 32), src);
      const uint32_t a2 = (uint32_t)__shfl((int)(uint32_t)noff, src);
      const uint32_t b0 = (uint32_t)__shfl((int)(uint32_t)rid, src), b1 = (uint32_t)__shfl((int)(uint32_t)(rid >> 32), src);
      const uint32_t b2 = (uint32_t)__shfl((int)si.y, src);
      dst[64 * h + lane] = second ? make_ulonglong2((uint64_t)b0 | ((uint64_t)b1 << 32), (uint64_t)b2)
                                  : make_ulonglong2((uint64_t)a0 | ((uint64_t)a1 << 32), (uint64_t)a2);
    }
    woff += 64;
    return;
  }
  uint32_t incl = cnt;
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= (uint32_t)d) incl += t;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63);
  if (total == 0) return;
  if (!__any(cnt > 2)) {
    const uint64_t out0 = woff + (incl - cnt);
    for (uint32_t occ = 0; occ < 2; ++occ) {
      if (occ < cnt && out0 + occ < cap) {
        uint64_t nid, noff;
        resolve_hit(mv, ent, sh, occ, nid, noff);
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(hits + out0 + occ);
        dst[0] = make_ulonglong2(nid, noff);
        dst[1] = make_ulonglong2(rec_offset + si.x, (uint64_t)si.y);
      }
    }
  } else {
    for (uint32_t j = lane; j - lane < total; j += 64) {      // wave-uniform trip count
      uint32_t a = 0, b = 63;                                  // owner: first seed whose inclusive prefix exceeds j
      for (int it = 0; it < 6; ++it) {
        uint32_t mid = (a + b) >> 1;
        uint32_t v = (uint32_t)__shfl((int)incl, (int)mid);
        if (v > j) b = mid; else a = mid + 1;
      }
      const int o = (int)min(a, 63u);
      SeedHits oh;
      oh.lo = (uint32_t)__shfl((int)sh.lo, o); oh.con = (uint32_t)__shfl((int)sh.con, o);
      oh.aux = (uint32_t)__shfl((int)sh.aux, o); oh.on_node = (uint32_t)__shfl((int)sh.on_node, o);
      oh.on_noff = (uint32_t)__shfl((int)sh.on_noff, o); oh.ofirst = (uint32_t)__shfl((int)sh.ofirst, o);
      oh.ocnt = (uint32_t)__shfl((int)sh.ocnt, o); oh.onoff = (uint32_t)__shfl((int)sh.onoff, o);
      const uint32_t o_excl = (uint32_t)__shfl((int)(incl - cnt), o);
      const uint32_t o_rid = (uint32_t)__shfl((int)si.x, o), o_roff = (uint32_t)__shfl((int)si.y, o);
      if (j < total && woff + j < cap) {
        uint64_t nid, noff;
        resolve_hit(mv, ent, oh, j - o_excl, nid, noff);
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(hits + woff + j);
        dst[0] = make_ulonglong2(nid, noff);
        dst[1] = make_ulonglong2(rec_offset + o_rid, (uint64_t)o_roff);
      }
    }
  }
  woff += total;
}

// K2 for sa_rate == 1 (the whole suffix array is resident): no LF-walk, so no quad cooperation.
// A wave round takes 64 seeds, one per lane.  When no seed of the round has more than two hits
// (the usual case) every lane writes its own: the records of consecutive lanes are consecutive.
// Otherwise the round's HITS are handed out to the lanes 64 at a time: lane j finds the seed that
// owns hit j by bisecting the prefix of the counts (shuffles), so a seed with many occurrences
// is spread over the wave instead of serialising one lane.
__global__ void __launch_bounds__(256)
k_fm_locate_direct(MapView mv, SeedOut so, bool have_off, const LocusEnt* __restrict__ ent,
                   const uint64_t* __restrict__ wave_off, const uint64_t* __restrict__ params, uint64_t seeds_cap,
                   uint32_t per_wave, const uint2* __restrict__ seed_info, uint64_t rec_offset,
                   psigpu_hit* __restrict__ hits, uint64_t cap)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_items = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_items, s0 + per_wave);
  uint64_t woff = s0 < s1 ? wave_off[wave] : 0;
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t item = base + lane;
    const bool have = item < s1;
    SeedHits sh = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (have) {
      sh.lo = so.iv_lo[item]; sh.con = so.iv_cnt[item];
      if (have_off) sh.ocnt = so.off_cnt[item];
    }
    const uint32_t cnt = sh.con + (sh.ocnt & ~OFF_INLINE);     // on-path occurrences first, then the table's loci
    // the rest of a seed's description is only read for seeds that have hits
    uint2 si = make_uint2(0, 0);
    if (cnt) {
      si = seed_info[item];
      if (sh.con) {
        sh.aux = so.iv_aux[item];
        if (sh.aux & AUX_RESOLVED) { sh.on_node = so.on_node[item]; sh.on_noff = so.on_noff[item]; }
      }
      if (sh.ocnt) { sh.ofirst = so.off_first[item]; sh.onoff = so.off_noff[item]; }
    }
    emit_round(mv, ent, sh, cnt, si, woff, rec_offset, hits, cap);
  }
}

// ... and what the emission needs of a seed, from its look-up result (the 16-byte form); EXT: the record is read here
__device__ __forceinline__ SeedHits res_to_hits(const uint4 r, const KmerSlot* __restrict__ ext, bool want_on, bool want_off, uint32_t gocc_thr,
                                                bool counts_from_record)
{
  SeedHits sh = { 0, 0, 0, 0, 0, 0, 0, 0 };
  sh.con = r.z & RES_CNT;
  uint32_t coff = r.w & ~RES_INLINE;
  if (r.z & RES_EXT) {
    if (counts_from_record || (sh.con | coff)) {
      const uint4* e = reinterpret_cast<const uint4*>(ext + r.x);
      const uint4 e0 = e[0], e1 = e[1];               // key, on_a, on_b | off_a, off_b, on_cnt, off_cnt
      if (counts_from_record) { const uint2 cc = ext_counts(e1, want_on, want_off, gocc_thr); sh.con = cc.x; coff = cc.y; }
      if (sh.con) {
        if (e1.z & KT_INLINE) { sh.on_node = e0.z; sh.on_noff = e0.w; sh.aux = AUX_RESOLVED; }
        else { sh.lo = e0.z; sh.aux = AUX_ONPOS; }
      }
      if (coff) { sh.ofirst = e1.x; sh.onoff = e1.y; sh.ocnt = (e1.w & KT_INLINE) ? (1u | OFF_INLINE) : coff; }
    }
  } else {
    if (sh.con) { sh.on_node = r.x; sh.on_noff = r.y; sh.aux = AUX_RESOLVED; }
    if (coff) { sh.ofirst = r.x; sh.onoff = r.y; sh.ocnt = 1u | OFF_INLINE; }
  }
  return sh;
}

// K2 of the k-mer table mode: a stream.  The probe left 16 bytes per seed (k_kmer_probe); this
// kernel turns them into records at the scan-given offsets.  EMIT_G rounds of 64 seeds are
// requested together, then emitted one after the other in seed order, with the same two paths as
// k_fm_locate_direct: own hits per lane when no seed of the round has more than two, hits handed
// out to the lanes otherwise.
constexpr int EMIT_G = 2;

template <bool R8>
__global__ void __launch_bounds__(256)
k_kmer_emit(MapView mv, const uint4* __restrict__ seed_res, const KmerSlot* __restrict__ ext,
            const LocusEnt* __restrict__ ent, const uint64_t* __restrict__ wave_total,
            const uint64_t* __restrict__ wave_total_off, const uint64_t* __restrict__ params,
            uint64_t seeds_cap, uint32_t per_wave, const uint2* __restrict__ seed_info, uint64_t rec_offset,
            psigpu_hit* __restrict__ hits, uint64_t cap, DevCounters* ctr, bool want_on, bool want_off, uint32_t gocc_thr,
            uint32_t uni_spr = 0, uint32_t uni_step = 0 /* seed_info == nullptr: seed s is seed s % spr of read s / spr */,
            bool plain_stores = false /* A/B: PSIGPU_PLAIN_STORES */)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_items = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_items, s0 + per_wave);
  // First output slot of this wave = hits of all the waves before it.  The per-wave totals of the
  // probe (8192 x 2 values, L2-resident) are summed here, by every workgroup for itself, instead of
  // by a kernel of their own between the probe and this one; the last workgroup leaves the totals.
  __shared__ uint64_t s_all[4], s_on[4];
  const uint32_t wib = threadIdx.x >> 6, w_first = blockIdx.x * 4;
  {
    uint64_t p_all = 0, p_on = 0;
    for (uint32_t i = threadIdx.x; i < w_first; i += 256) {
      const uint64_t a = wave_total[i];
      p_on += a; p_all += a + wave_total_off[i];
    }
    for (int d = 32; d > 0; d >>= 1) { p_all += __shfl_down(p_all, d); p_on += __shfl_down(p_on, d); }
    if (lane == 0) { s_all[wib] = p_all; s_on[wib] = p_on; }
  }
  __syncthreads();
  uint64_t woff = s_all[0] + s_all[1] + s_all[2] + s_all[3];
  uint64_t on_before = s_on[0] + s_on[1] + s_on[2] + s_on[3];
  for (uint32_t w = 0; w < 4; ++w) {
    const uint64_t a = wave_total[w_first + w], b = wave_total_off[w_first + w];
    if (w < wib) woff += a + b;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
      on_before += a;
      if (w == 3) {
        uint64_t all = s_all[0] + s_all[1] + s_all[2] + s_all[3];
        for (uint32_t x = 0; x < 4; ++x) all += wave_total[w_first + x] + wave_total_off[w_first + x];
        ctr->n_hits_on.v = on_before;
        ctr->n_hits_tab.v = all;
      }
    }
  }
  for (uint64_t base = s0; base < s1; base += 64 * EMIT_G) {
    uint4 rr[EMIT_G];
    uint2 ss[EMIT_G];
#pragma unroll
    for (int g = 0; g < EMIT_G; ++g) {
      const uint64_t item = base + (uint64_t)g * 64 + lane;
      rr[g] = make_uint4(0, 0, 0, 0); ss[g] = make_uint2(0, 0);
      if (item < s1) {
        if constexpr (R8) {
          const uint64_t r8 = reinterpret_cast<const uint64_t*>(seed_res)[item];
          // (the 16-byte form of the same answer; a record's counts are taken from the record below)
          rr[g] = (r8 & R8_EXT) ? make_uint4((uint32_t)r8, 0u, RES_EXT, 0u)
                                : make_uint4((uint32_t)r8, (uint32_t)(r8 >> 32) & ((1u << R8_NOFF_BITS) - 1u),
                                             (r8 & R8_ON) ? (1u | RES_INLINE) : 0u, (r8 & R8_OFF) ? (1u | RES_INLINE) : 0u);
        } else rr[g] = seed_res[item];
        if (seed_info) ss[g] = seed_info[item];
        else { const uint32_t rd = (uint32_t)item / uni_spr; ss[g] = make_uint2(rd, ((uint32_t)item - rd * uni_spr) * uni_step); }
      }
    }
#pragma unroll
    for (int g = 0; g < EMIT_G; ++g) {
      const uint2 si = ss[g];
      const SeedHits sh = res_to_hits(rr[g], ext, want_on, want_off, gocc_thr, R8);
      const uint32_t cnt = sh.con + (sh.ocnt & ~OFF_INLINE);     // on-path occurrences first, then the loci
      emit_round(mv, ent, sh, cnt, si, woff, rec_offset, hits, cap, !plain_stores);
    }
  }
}

// ------------------------------------------------------------------------------------
// The default step in ONE kernel (round 5): seeding, the k-mer table probe and the emission of a TILE of seeds by one
// workgroup, where rounds 1-4 ran k_seed_pack -> k_kmer_probe -> k_kmer_emit with 8 bytes of key and 8 bytes of result per
// seed written and read back in between (a third of the step's traffic, two launches, and a probe kernel with one load
// in flight per lane).  A workgroup takes the next tile (KS_R rounds of 256 seeds; a ticket, so tiles start in order),
// packs its seeds' keys in registers, issues the first table load of all its rounds before it looks at any, counts the
// tile's hits and learns its first output slot by a decoupled look-back over the tiles before it: tile_state[t] is ONE
// 64-bit word -- flag (aggregate / inclusive prefix), the call's serial number, the count -- so a word is either this
// call's or ignored and nothing needs a fence.  The records come out in seed order exactly as k_kmer_emit writes them
// (emit_round: the transposed stores, the spread of a seed with many hits over the wave).
// The general (not equal-length) reads locate their read as k_seed_pack does, from the scanned seed offsets.
// ------------------------------------------------------------------------------------
constexpr int KS_R = 4;
constexpr uint32_t KS_TILE = 256 * KS_R;
constexpr uint64_t KS_AGG = 1ull << 62, KS_PFX = 2ull << 62, KS_VAL = (1ull << 40) - 1;
constexpr uint32_t KS_SERIAL = (1u << 22) - 1;
__device__ __forceinline__ uint64_t ks_word(uint64_t flag, uint32_t serial22, uint64_t v) { return flag | ((uint64_t)serial22 << 40) | v; }

template <bool PACKED, bool UNIFORM>
__global__ void __launch_bounds__(256)
k_kmer_step(const char* __restrict__ bases, const uint64_t* __restrict__ read_off, const uint64_t* __restrict__ seed_off, uint64_t n_reads,
            const uint64_t* __restrict__ params, uint64_t seeds_cap, uint64_t n_bases, uint32_t k, uint32_t step, PackedIn pk, UniformIn un,
            KmerTableView kt, MapView mv, const LocusEnt* __restrict__ ent, bool want_on, bool want_off, uint32_t gocc_thr,
            uint64_t rec_offset, psigpu_hit* __restrict__ hits, uint64_t cap, uint64_t* tile_state, uint32_t serial22,
            DevCounters* ctr, bool plain_stores)
{
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_cnt[KS_R * 4];
  __shared__ uint64_t s_prefix;
  const uint32_t lane = lane_id(), wib = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = (uint32_t)atomicAdd(&ctr->ticket.v, 1ull);
  __syncthreads();
  const uint64_t tile = s_tile;
  const uint64_t n_seeds = min(params[0], seeds_cap), ratio = params[1];
  const uint64_t t0 = tile * KS_TILE;
  if (t0 >= n_seeds) return;                      // (the grid is sized by the upper bound of the seed count)

  // ---- the seeds of this lane: read, offset in the read, key ----------------------------------------------------
  uint64_t key[KS_R];
  uint2 si[KS_R];
  uint32_t nok = 0;
  const uint32_t nw = (k + 7) >> 3;
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    const uint64_t s = t0 + (uint64_t)r * 256 + threadIdx.x;
    key[r] = KEY_INVALID; si[r] = make_uint2(0, 0);
    if (s >= n_seeds) continue;
    uint64_t rd, so0, ro;
    if constexpr (UNIFORM) {
      rd = (uint32_t)s / un.spr; so0 = rd * un.spr; ro = rd * un.len;
      // the claim, checked where it is used: this read starts and ends where equal lengths put it
      if (read_off[rd] != ro || read_off[rd + 1] != ro + un.len) ctr->not_uniform.v = 1ull;
    } else {
      rd = __umul64hi(s, ratio);
      if (rd >= n_reads) rd = n_reads - 1;
      so0 = seed_off[rd];
      const uint64_t so1 = seed_off[rd + 1];
      if (!(so0 <= s && s < so1)) {               // wrong guess (ragged reads): gallop to a bracket, bisect
        uint64_t l = rd, hi;
        if (so0 <= s) {
          uint64_t d = 1;
          while (l + d < n_reads && seed_off[l + d] <= s) { l += d; d <<= 1; }
          hi = min(l + d, n_reads);
        } else {
          uint64_t d = 1;
          hi = l;
          while (d < hi && seed_off[hi - d] > s) { hi -= d; d <<= 1; }
          l = d < hi ? hi - d : 0;
        }
        while (hi - l > 1) {                      // invariant: seed_off[l] <= s < seed_off[hi]
          const uint64_t mid = (l + hi) >> 1;
          if (seed_off[mid] <= s) l = mid; else hi = mid;
        }
        rd = l; so0 = seed_off[l];
      }
      ro = read_off[rd];
    }
    const uint64_t st = (s - so0) * step;
    si[r] = make_uint2((uint32_t)rd, (uint32_t)st);
    uint64_t kk = 0;
    uint32_t ok = 1;
    if constexpr (PACKED) {
      const uint64_t* __restrict__ P = reinterpret_cast<const uint64_t*>(bases);
      const uint64_t q = ro + st + pk.bias2;
      const uint64_t w = q >> 5;
      const uint32_t sh = 2u * (uint32_t)(q & 31);
      const uint64_t w0 = P[w], w1 = P[w + 1];    // (the buffer is padded: the word behind the window is loaded, none of its bits used)
      kk = (sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0) >> (64 - 2 * k);
      if (pk.mask) {
        const uint64_t qm = ro + st + pk.biasm;
        const uint32_t ms = (uint32_t)(qm & 63);
        const uint64_t m0 = pk.mask[qm >> 6], m1 = pk.mask[(qm >> 6) + 1];
        const uint64_t win = ms ? (m0 >> ms) | (m1 << (64 - ms)) : m0;
        ok = (win & ((1ull << k) - 1ull)) == 0;
      }
    } else {
      const uint64_t abs0 = ro + st;
      if (abs0 + 8ull * nw <= n_bases) {
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w)
          if (w < nw) {
            uint64_t x;
            __builtin_memcpy(&x, bases + abs0 + 8 * w, 8);
            const uint32_t take = min(8u, k - 8 * w);
            kk = (kk << (2 * take)) | pack8(x, take, ok);
          }
      } else {
        const char* p = bases + abs0;
        for (uint32_t i = 0; i < k; ++i) {        // tail of the buffer: byte loads
          int b = base2(p[i]);
          if (b < 0) { ok = 0; b = 0; }
          kk = (kk << 2) | (uint64_t)b;
        }
      }
    }
    key[r] = ok ? kk : KEY_INVALID;
    nok += ok;
  }

  // ---- one probe per seed: every round's first load in flight before the first is looked at ----------------------------
  uint64_t h[KS_R];
  uint4 v[KS_R];
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    h[r] = 0; v[r] = make_uint4(0, 0, NIL, NIL);
    if (key[r] != KEY_INVALID) { h[r] = kt_home(key[r], kt.n_slots); v[r] = load16(kt.ht + h[r]); }
  }
  uint4 res[KS_R];
  uint32_t on_sum = 0, n_live = 0;
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    res[r] = make_uint4(0, 0, 0, 0);
    if (key[r] != KEY_INVALID) res[r] = kt_resolve(kt, key[r], h[r], v[r], want_on, want_off, gocc_thr);
    const uint32_t con = res[r].z & RES_CNT, coff = res[r].w & ~RES_INLINE;
    on_sum += con; n_live += con != 0;
    uint32_t c = con + coff;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) s_cnt[r * 4 + wib] = c;
  }
  for (int d = 32; d > 0; d >>= 1) { nok += __shfl_down(nok, d); on_sum += __shfl_down(on_sum, d); n_live += __shfl_down(n_live, d); }
  if (lane == 0) {
    if (nok) ctr->n_seeds_valid.add((unsigned long long)nok);
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
    if (on_sum) ctr->n_hits_on_s.add((unsigned long long)on_sum);
  }
  __syncthreads();

  // ---- first output slot of the tile: decoupled look-back (wave 0) ---------------------------------------------------
  if (wib == 0) {
    uint64_t agg = 0;
#pragma unroll
    for (int i = 0; i < KS_R * 4; ++i) agg += s_cnt[i];
    uint64_t excl = 0;
    if (tile != 0) {
      if (lane == 0) __hip_atomic_store(&tile_state[tile], ks_word(KS_AGG, serial22, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t top = (int64_t)tile - 1;              // the window: tiles top, top - 1, ... top - 63 on lanes 0 .. 63
      while (true) {
        const int64_t idx = top - (int64_t)lane;
        // (before tile 0: an inclusive prefix of nothing)
        const uint64_t w = idx >= 0 ? __hip_atomic_load(&tile_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ks_word(KS_PFX, serial22, 0);
        const bool ready = (w >> 62) != 0 && (uint32_t)((w >> 40) & KS_SERIAL) == serial22;
        const uint64_t m_ready = __ballot(ready), m_pfx = __ballot(ready && (w >> 62) == 2);
        const uint32_t n_ready = m_ready == ~0ull ? 64u : (uint32_t)__ffsll((long long)~m_ready) - 1u;      // tiles ready from the window's top
        const uint32_t first_pfx = m_pfx ? (uint32_t)__ffsll((long long)m_pfx) - 1u : 64u;
        if (first_pfx < n_ready || (first_pfx == 64u && n_ready == 64u)) {
          const uint32_t upto = first_pfx < 64u ? first_pfx : 63u;      // add lanes 0 .. upto
          uint64_t part = lane <= upto ? (w & KS_VAL) : 0ull;
          for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
          excl += part;
          if (first_pfx < 64u) break;
          top -= 64;
        } else __builtin_amdgcn_s_sleep(2);
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&tile_state[tile], ks_word(KS_PFX, serial22, excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = excl;
      if (t0 + KS_TILE >= n_seeds) ctr->n_hits_tab.v = excl + agg;      // the last tile: what the step wrote (or would have, past cap)
    }
  }
  __syncthreads();

  // ---- records ---------------------------------------------------------------------------------------------------
  uint64_t woff = s_prefix;
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    uint64_t mine = woff;
    for (uint32_t w = 0; w < wib; ++w) mine += s_cnt[r * 4 + w];
    const SeedHits sh = res_to_hits(res[r], kt.ext, want_on, want_off, gocc_thr, false);
    const uint32_t cnt = sh.con + (sh.ocnt & ~OFF_INLINE);     // on-path occurrences first, then the loci
    emit_round(mv, ent, sh, cnt, si[r], mine, rec_offset, hits, cap, !plain_stores);
    woff += (uint64_t)s_cnt[r * 4] + s_cnt[r * 4 + 1] + s_cnt[r * 4 + 2] + s_cnt[r * 4 + 3];
  }
}

// ------------------------------------------------------------------------------------
// K4: traverser.  One wavefront per workgroup; each wave owns a contiguous chunk of
// starting loci and an LDS stack of partial walks.  Every iteration each lane pops one
// partial walk (or takes a fresh locus), extends it through one node, and either
// completes (probe the seed table, emit) or forks one partial walk per out-edge, pushed
// with wave ballot + prefix counts.  Walks die on N, at sinks before k bases
// (traverser_bfs.hpp:124,141-144).  Items that do not fit the LDS stack go to a global spill
// queue that is drained by re-launching the kernel on it.
// ------------------------------------------------------------------------------------
#ifndef TRAV_CAP_N
#define TRAV_CAP_N 128
#endif
#ifndef TRAV_WIN_N
#define TRAV_WIN_N 256
#endif
constexpr uint32_t TRAV_WIN = TRAV_WIN_N;   // node records staged in LDS per wave (16 B each)
constexpr int TRAV_CAP = TRAV_CAP_N;   // LDS stack entries per wave (16 B each)

struct GraphView {
  const NodeRec* nodes;
  const NodeLite* lite;
  const uint64_t* lab2;      // 2-bit bases, 32 per word, first base most significant
  const uint64_t* labn;      // N mask, 64 per word, first base most significant
  const uint32_t* edge_to;
  const uint64_t* node_id;      // rank -> external id ...
  uint64_t id_base;             // ... or id = rank + id_base when the ids are consecutive (no load)
  bool id_affine;
};

struct TableView {
  const TableSlot* ht;                            // every bucket's region one after the other (k_sb_build)
  const uint64_t* boff; uint32_t n_wg, pb;        // bucket b: slots [2 boff[b n_wg], 2 boff[(b + 1) n_wg])
  const uint32_t* seed_next; const uint2* seed_info;
  const void* seed_wide;                          // two-word seeds: the k-mer of every seed (u128), else nullptr
  const uint32_t* pfx12;                          // 4^12-bit prefix bitmap (nullptr when k < 12)
  const uint32_t* pfx_bits; uint32_t pfx_len;     // prefix bitmap of the seeds, 4^pfx_len bits
};

// `cnt` (1..32) bases starting at base index `at`, right-aligned
__device__ __forceinline__ uint64_t fetch_bases(const uint64_t* lab2, uint64_t at, uint32_t cnt)
{
  uint64_t w = at >> 5; uint32_t sh = (uint32_t)(at & 31) * 2;
  uint64_t x = lab2[w] << sh;
  if (sh + 2 * cnt > 64) x |= lab2[w + 1] >> (64 - sh);
  return x >> (64 - 2 * cnt);
}

__device__ __forceinline__ bool any_n(const uint64_t* labn, uint64_t at, uint32_t cnt)
{
  uint64_t w = at >> 6; uint32_t sh = (uint32_t)(at & 63);
  uint64_t x = labn[w] << sh;
  if (sh + cnt > 64) x |= labn[w + 1] >> (64 - sh);
  return (x >> (64 - cnt)) != 0;
}

// Every lane runs a depth-first walk of its own: it extends its partial walk through one
// node per iteration, continues in place along the first out-edge and pushes one partial
// walk per further out-edge on the wave's LDS stack (the reference does the same on its
// state vector: first edge in place, copies for the others, traverser_bfs.hpp:146-160).
// Idle lanes pop from the stack, then take fresh loci from an LDS buffer that is refilled 64
// loci at a time from loads issued one refill earlier.  A walk whose first 12 / 14 bases are
// the prefix of no seed is dropped: "a base with no continuation in the seeds index"
// (traverser_bfs.hpp:124) -- the reads-index descent of the reference restated as bitmap
// probes.  Complete walks are queued in LDS and looked up in the seed table 64 at a time, so
// the walking loop carries two dependent global loads per iteration (node record, bitmaps)
// and the table / emit chain is paid once per 64 k-mers.
constexpr int DONE_CAP = 128;          // completed k-mers waiting for the table lookup

template <typename KEY> struct DoneItemT { KEY kmer; uint32_t locus; uint32_t pad; };

template <typename KEY>
__device__ __forceinline__ void
process_done(const GraphView& g, const TableView& tb, const uint2* __restrict__ loci, const DoneItemT<KEY>* dq, uint32_t n, uint32_t k,
             uint64_t rec_offset, ChunkWriter& cw, DevCounters* ctr)
{
  // lanes 0..n-1 take one completed k-mer each
  const uint32_t lane = lane_id();
  uint32_t s = NIL, dup = NIL, locus = 0;
  KEY want = 0;
  if (lane < n) {
    DoneItemT<KEY> d = dq[lane];
    want = d.kmer;
    const uint64_t tkey = table_key(d.kmer);
    locus = d.locus;
    const uint32_t b = sb_bucket(tkey, k, tb.pb);
    const uint32_t lo = (uint32_t)tb.boff[(uint64_t)b * tb.n_wg], m = 2 * ((uint32_t)tb.boff[(uint64_t)(b + 1) * tb.n_wg] - lo);
    if (m) {
      const TableSlot* region = tb.ht + 2ull * lo;
      uint32_t h = sb_home(tkey, m);
      while (true) {
        const uint4 raw = load16(region + h);
        TableSlot sl = { (unsigned long long)raw.x | ((unsigned long long)raw.y << 32), raw.z, raw.w };
        if (sl.key == tkey) { s = sl.val; dup = sl.dup; break; }
        if (sl.key == KEY_INVALID) break;
        h = h + 1 < m ? h + 1 : 0;
      }
    }
  }
  if (!__any(s != NIL)) return;
  uint64_t nid = 0, noff = 0;
  if (s != NIL) { uint2 lc = loci[locus]; nid = g.id_affine ? g.id_base + lc.x : g.node_id[lc.x]; noff = lc.y; }
  while (__any(s != NIL)) {
    bool has = s != NIL;
    uint64_t rid = 0, roff = 0;
    uint32_t nx = NIL;
    if (has) {
      // (two-word seeds: the table is keyed by a fingerprint -- a seed counts only when its k-mer is the walk's)
      if constexpr (sizeof(KEY) > 8) has = reinterpret_cast<const u128*>(tb.seed_wide)[s] == want;
      uint2 si = tb.seed_info[s]; rid = rec_offset + si.x; roff = si.y;
      nx = dup;                                   // then down the duplicate chain
      if (dup != NIL) dup = tb.seed_next[dup];
    }
    chunk_emit(cw, has, nid, noff, rid, roff, ctr);
    s = nx;
  }
}

template <bool ENUM, typename KEY = uint64_t>
__global__ void __launch_bounds__(64)
k_traverse(GraphView g, TableView tb, const uint2* __restrict__ loci /* (node rank, offset) */,
           uint64_t n_loci, uint32_t loci_per_wave,
           const TravItemT<KEY>* __restrict__ spill_in, uint64_t n_spill_in,
           TravItemT<KEY>* __restrict__ spill_out, uint64_t spill_cap,
           uint32_t k, uint64_t rec_offset, psigpu_hit* __restrict__ chunks, uint32_t* __restrict__ chunk_fill,
           uint32_t cap_chunks, uint64_t n_nodes, DevCounters* ctr, EnumOut eo, const uint4* __restrict__ pfx_roots = nullptr)
{
  // pfx_roots (round 4, query time, k > 12): the roots are not the loci but their PREFIX WALKS, enumerated once per index
  // (ensure_pfx_roots): (12-mer, node, offset in the node's record, locus) -- where a walk from the locus stands after 12
  // bases.  What every chunk did for every locus -- load the locus, load its node record, hop to the next node for the rest
  // of the 12 bases -- is then a coalesced stream of 16-byte records, checked against the chunk's 12-mer map (L2-resident)
  // while it is staged: only the third or so of the walks the map lets pass ever enters the walking loop.
  typedef TravItemT<KEY> TravItem;
  typedef DoneItemT<KEY> DoneItem;
  static_assert(!ENUM || sizeof(KEY) == 8, "the tables are made for one-word seeds");
  __shared__ TravItem stack[TRAV_CAP];
  __shared__ DoneItem doneq[DONE_CAP];
  __shared__ TravItem rootbuf[64];        // staged roots and their start offsets
  __shared__ uint32_t rootoff[64];
  __shared__ NodeLite window[TRAV_WIN];   // node records of the ranks this wave's loci start in
  ChunkWriter cw = { chunks, chunk_fill, cap_chunks, NIL, 0 };
  PairWriter pw = { NIL, 0 };
  const uint32_t lane = lane_id();
  // roots: either fresh loci (spill_in == nullptr) or spilled partial walks
  const bool from_spill = spill_in != nullptr;
  const bool from_pfx = !ENUM && !from_spill && pfx_roots != nullptr;      // (n_loci then counts prefix walks)
  const uint64_t n_roots = from_spill ? n_spill_in : n_loci;
  uint64_t cursor = (uint64_t)blockIdx.x * loci_per_wave;     // next root NOT yet requested from memory
  const uint64_t cend = min(n_roots, cursor + loci_per_wave);
  uint32_t top = 0, ndone = 0;            // wave-uniform
  uint32_t rb_pos = 0, rb_cnt = 0;        // wave-uniform: staged roots [rb_pos, rb_cnt) are unread
  uint32_t kpaths = 0;
#ifdef TRAV_STATS
  uint32_t dbg_iters = 0, dbg_lanes = 0;
#endif
  // prefetch registers: this lane's root of the NEXT refill
  TravItem pf = { 0, 0, 0 };
  uint32_t pf_off = 0;
  uint32_t pf_cnt = 0;                    // wave-uniform: roots held in the prefetch registers
  bool pf_keep = false;                   // prefix roots: this lane's prefetched walk passes the chunk's 12-mer map
  auto prefetch = [&]() {
    pf_cnt = (uint32_t)min((uint64_t)64, cend - cursor);
    pf_keep = lane < pf_cnt;
    if (lane < pf_cnt) {
      uint64_t rix = cursor + lane;
      if (from_spill) { pf = spill_in[rix]; pf_off = 0; }
      else if (from_pfx) {
        const uint4 e = pfx_roots[rix];       // 12-mer, node, offset, locus
        pf.kmer = (KEY)e.x | ((KEY)1 << (2 * PFX_SHORT)); pf.node = e.y; pf.locus = e.w; pf_off = e.z;
        if (tb.pfx12) pf_keep = (tb.pfx12[e.x >> 5] >> (e.x & 31)) & 1u;
      }
      else { uint2 lc = loci[rix]; pf.kmer = 1; pf.node = lc.x; pf.locus = (uint32_t)rix; pf_off = lc.y; }
    }
    cursor += pf_cnt;
  };
  // The loci of a wave are consecutive, so are the ranks of the nodes they start in, and (for
  // graphs whose ranks follow the topology, as vg's do) so are the nodes the walks hop to: stage
  // that rank window in LDS once, coalesced; anything outside is read from memory.
  uint32_t wb = 0, win_n = 0;             // first rank / size of the window (none for spill launches)
  if (!from_spill && cursor < cend) {
    wb = from_pfx ? pfx_roots[cursor].y : loci[cursor].x;
    win_n = TRAV_WIN;
    for (uint32_t i = lane; i < TRAV_WIN; i += 64) {
      NodeLite z = { 0, NIL, LITE_SLOW };
      window[i] = (uint64_t)wb + i < n_nodes ? g.lite[(uint64_t)wb + i] : z;
    }
  }
  prefetch();
  bool have = false;
  TravItem it = { 0, 0, 0 };
  uint32_t off = 0;

  while (true) {
    // ---- idle lanes: pop a pending fork, else take a staged root ---------------------------
    uint64_t nm = __ballot(!have);
    if (nm) {
      uint32_t nneed = (uint32_t)__popcll(nm), myr = (uint32_t)__popcll(nm & lanemask_lt());
      uint32_t npop = min(top, nneed);
      while (nneed > npop && rb_pos == rb_cnt && pf_cnt) {
        // refill the staged roots from the prefetch registers and start the next prefetch (prefix roots: only the
        // walks the 12-mer map lets pass are staged -- possibly none of a refill, hence the loop)
        const uint64_t km = __ballot(pf_keep);
        if (pf_keep) { const uint32_t at = (uint32_t)__popcll(km & lanemask_lt()); rootbuf[at] = pf; rootoff[at] = pf_off; }
        rb_pos = 0; rb_cnt = (uint32_t)__popcll(km);
        prefetch();
        __builtin_amdgcn_wave_barrier();
      }
      uint32_t nroot = min(nneed - npop, rb_cnt - rb_pos);
      if (!have) {
        if (myr < npop) { it = stack[top - 1 - myr]; off = 0; have = true; }
        else if (myr - npop < nroot) { it = rootbuf[rb_pos + myr - npop]; off = rootoff[rb_pos + myr - npop]; have = true; }
      }
      top -= npop;
      rb_pos += nroot;
    }
    if (!__any(have)) break;              // stack, staged roots and prefetch are all empty
    __builtin_amdgcn_wave_barrier();
#ifdef TRAV_STATS
    ++dbg_iters; dbg_lanes += (uint32_t)__popcll(__ballot(have));
#endif

    // ---- extend through one node ---------------------------------------------------
    uint32_t nchild = 0, e_off = 0, end_off = 0;
    KEY fork_kmer = 0;
    bool done = false;
    if (have) {
      uint32_t widx = it.node - wb;                                     // wraps above the window
      uint4 nlw;                                                        // (one 16-byte read from either place)
      if (widx < win_n) nlw = *reinterpret_cast<const uint4*>(&window[widx]); else nlw = *reinterpret_cast<const uint4*>(&g.lite[it.node]);
      keep_whole(nlw);
      NodeLite nl = { (uint64_t)nlw.x | ((uint64_t)nlw.y << 32), nlw.z, nlw.w };
      uint32_t depth = hibit(it.kmer) >> 1;
      KEY b = 0;
      uint32_t take, e1 = 0, coff = 0;
      bool dead = false;
      if (!(nl.meta & LITE_SLOW)) {
        uint32_t len = nl.meta & 63u;
        take = min(k - depth, len - off);
        if (take) b = (nl.head2 << (2 * off)) >> (64 - 2 * take);
        nchild = (nl.meta >> 6) & 3u;
        coff = (nl.meta >> 8) & 63u;
        e1 = nl.edge0 + (uint32_t)((int32_t)nl.meta >> 16);
        e_off = e1;                                                     // only read when nchild == 2
      } else {
        NodeRec nr = g.nodes[it.node];
        take = min(k - depth, nr.len - off);
        if (take) {
          if (!((nr.w0 >> 62) & 1)) {         // extended head in the full record
            dead = ((nr.headn << off) >> (32 - take)) != 0;
            b = (nr.head2 << (2 * off)) >> (64 - 2 * take);
          } else {                            // long node: label words
            uint64_t lab = nr.w0 & 0xFFFFFFFFFFull;
            dead = (nr.w0 >> 63) && any_n(g.labn, lab + off, take);
            if (sizeof(KEY) > 8 && take > 32)           // (two-word seeds: up to 63 bases of one long node at a time)
              b = dead ? (KEY)0 : (((KEY)fetch_bases(g.lab2, lab + off, 32) << (2 * (take - 32))) |
                                   (KEY)fetch_bases(g.lab2, lab + off + 32, take - 32));
            else
            b = dead ? 0 : fetch_bases(g.lab2, lab + off, take);
          }
        }
        nchild = (uint32_t)(nr.w0 >> 40) & 0xFFFFu;
        coff = (uint32_t)(nr.w0 >> 56) & 63u;
        e_off = nr.edge_off;
        nl.edge0 = nr.edge0;
      }
      end_off = off + take;               // where the walk stands in this node's record after the bases it took
      if (take && !dead) {
        KEY body = it.kmer ^ ((KEY)1 << (2 * depth));
        body = (body << (2 * take)) | b;
        uint32_t nd = depth + take;
        // seed-prefix filter, once per level, when the walk first reaches that many bases.  The long map is
        // only asked when the short one (2 MiB, L2-resident) lets the walk pass: the kernel runs at the fabric's
        // request rate, not at a latency, and a probe of the 32-MiB map is a request that leaves L2
        bool c12 = tb.pfx12 && depth < PFX_SHORT && nd >= PFX_SHORT;
        bool c14 = tb.pfx_bits && depth < tb.pfx_len && nd >= tb.pfx_len;
        uint32_t w12 = 0xFFFFFFFFu, w14 = 0xFFFFFFFFu, p12 = 0, p14 = 0;
        if (c12) { p12 = (uint32_t)(body >> (2 * (nd - PFX_SHORT))); w12 = tb.pfx12[p12 >> 5]; }
        const bool pass12 = (w12 >> (p12 & 31)) & 1u;
        if (c14 && pass12) { p14 = (uint32_t)(body >> (2 * (nd - tb.pfx_len))); w14 = tb.pfx_bits[p14 >> 5]; }
        dead = !(pass12 && ((w14 >> (p14 & 31)) & 1u));
        depth = nd;
        it.kmer = body | ((KEY)1 << (2 * depth));
      }
      if (dead) { have = false; nchild = 0; }
      else if (depth == k) { done = true; have = false; nchild = 0; }
      else {
        fork_kmer = it.kmer;
        if (nchild == 0) have = false;    // sink before k bases (traverser_bfs.hpp:141-144)
        else { it.node = nl.edge0; off = coff; }
      }
    }

    // ---- complete walks: queue the k-mer; look the queue up in the seed table 64 at a time ----
    if constexpr (ENUM) {
      // table construction: every complete walk of a locus is recorded, up to walk_cap per locus;
      // a locus that went over the cap stops forking (it is left to the query-time traverser)
      bool has = false;
      if (done) { ++kpaths; has = atomicAdd(&eo.walks[it.locus], 1u) < eo.walk_cap; }
      const uint64_t km_ = (uint64_t)(it.kmer ^ ((KEY)1 << (2 * k)));
      if (eo.prefix) pair_emit(eo, pw, has, km_ | ((uint64_t)it.node << 32), (uint64_t)it.locus | ((uint64_t)end_off << 32), ctr);
      else
      pair_emit(eo, pw, has, km_, it.locus, ctr);
      if (nchild > 1 && eo.walks[it.locus] > eo.walk_cap) { nchild = 0; have = false; }
    } else {
      uint64_t dm = __ballot(done);
      if (dm) {
        if (done) {
          ++kpaths;
          DoneItem d = { it.kmer ^ ((KEY)1 << (2 * k)), it.locus, 0 };
          doneq[ndone + (uint32_t)__popcll(dm & lanemask_lt())] = d;
        }
        ndone += (uint32_t)__popcll(dm);
        __builtin_amdgcn_wave_barrier();
        if (ndone >= 64) {
          process_done<KEY>(g, tb, loci, doneq + (ndone - 64), 64, k, rec_offset, cw, ctr);
          ndone -= 64;
        }
      }
    }

    // ---- fork: first out-edge continues in this lane, the others are pushed ----------------
    uint64_t fm = __ballot(nchild > 1);
    for (uint32_t j = 1; fm; ++j) {
      bool p = j < nchild;
      uint32_t slot = top + (uint32_t)__popcll(fm & lanemask_lt());
      if (p) {
        uint32_t tgt = (nchild == 2) ? e_off : g.edge_to[e_off + j];
        TravItem c = { fork_kmer, tgt, it.locus };
        if (slot < (uint32_t)TRAV_CAP) stack[slot] = c;
        else {
          unsigned long long q = atomicAdd(&ctr->n_spill.v, 1ull);
          if (q < spill_cap) spill_out[q] = c;
        }
      }
      top = min(top + (uint32_t)__popcll(fm), (uint32_t)TRAV_CAP);
      fm = __ballot(j + 1 < nchild);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (ENUM) {
    if (pw.id != NIL && pw.id < eo.cap_chunks && lane == 0) eo.fill[pw.id] = pw.n;
  } else {
    if (ndone) process_done<KEY>(g, tb, loci, doneq, ndone, k, rec_offset, cw, ctr);
    chunk_close(cw);
  }
#ifdef TRAV_STATS
  if (lane == 0) { atomicAdd(&ctr->dbg0.v, (unsigned long long)dbg_iters); atomicAdd(&ctr->dbg1.v, (unsigned long long)dbg_lanes); }
#endif
  for (int d = 32; d > 0; d >>= 1) kpaths += __shfl_down(kpaths, d);
  if (lane == 0 && kpaths) ctr->n_kpaths.add((unsigned long long)kpaths);
}

// ------------------------------------------------------------------------------------
// MEM mode: SeedFinder::seeds_on_paths( sequence, callback ) -> find_mems
// (reference include/psi/seed_finder.hpp:1459-1479, include/psi/index_iter.hpp:854-906).
// Per read, the reference walks its path-index iterator FORWARD through the read: from `start` it
// appends bases while the pattern still occurs on the indexed paths; the first time the pattern is
// at least `minlen` long and has at most gocc_threshold occurrences, every occurrence is a hit
// (read_offset = start, match_len = the pattern length, gocc = the number of occurrences) and the
// search starts again one base behind the end of the pattern; a base that cannot be appended (or an
// N) also restarts it, one base behind that base.  max_mem stops a read once that many hits are out.
//
// The reference can append because it indexes the REVERSED text; this index is over the forward text,
// whose FM half only prepends.  Appending is done on the suffix array instead: the rows whose suffix
// starts with the pattern are an interval, and the sub-interval whose next symbol is c is found by
// two bisections over (SA[row] + depth)-th text symbols (whole SA + 4-bit text resident: sa_rate 1).
// One lane per read: the walk is sequential inside a read and independent across reads.
// ------------------------------------------------------------------------------------
struct MemGroup { uint32_t read, start, plen, lo, cnt, part, total; };      // one reported pattern in one part: SA rows [lo, lo + cnt)

// suffix array, text and segment table of every part of the index (a pattern's occurrences are the union
// over the parts -- it never spans two paths -- and its occurrence count their sum)
struct MemPart { const uint32_t* sa; const uint64_t* text4; const SegRec* seg; const uint32_t* seg_dir; uint32_t n; };
struct MemParts { MemPart p[PSIGPU_MAX_PARTS]; uint32_t n_parts; };

__device__ __forceinline__ int text_sym(const uint64_t* __restrict__ text4, uint64_t n, uint64_t pos)
{
  if (pos >= n) return -1;
  const uint32_t nib = (uint32_t)(text4[pos >> 4] >> (60 - 4 * (pos & 15))) & 0xFu;
  return (nib & 4u) ? -1 : (int)(nib & 3u);        // separators / the sentinel sort in front of every base
}

// first row in [lo, hi) whose symbol at depth `d` is >= c  (rows of one interval are ordered by it)
__device__ __forceinline__ uint32_t mem_lower(const uint32_t* __restrict__ sa, const uint64_t* __restrict__ text4, uint64_t n,
                                              uint32_t lo, uint32_t hi, uint32_t d, int c)
{
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (text_sym(text4, n, (uint64_t)sa[mid] + d) < c) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(64)
k_find_mems(const char* __restrict__ bases, const uint64_t* __restrict__ read_off, uint64_t n_reads,
            MemParts mp, uint32_t minlen,
            uint32_t gocc_thr, uint32_t max_mem, MemGroup* __restrict__ groups, uint64_t cap_groups,
            unsigned long long* __restrict__ n_groups, unsigned long long* __restrict__ n_hits)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const char* pat = bases + read_off[r];
  const uint64_t len = read_off[r + 1] - read_off[r];
  uint64_t start = 0, nof = 0;
  uint32_t plen = 0, lo[PSIGPU_MAX_PARTS], hi[PSIGPU_MAX_PARTS];
#pragma unroll
  for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q) { lo[q] = 0; hi[q] = q < mp.n_parts ? mp.p[q].n : 0u; }
  bool has_hit = false;
  while (start + plen < len) {
    uint64_t total = 0;
#pragma unroll
    for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q) total += hi[q] - lo[q];
    if (plen >= minlen && total <= gocc_thr) {
      has_hit = true;
#pragma unroll
      for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q)
        if (hi[q] > lo[q]) {
          const unsigned long long g = atomicAdd(n_groups, 1ull);
          if (g < cap_groups) groups[g] = MemGroup{ (uint32_t)r, (uint32_t)start, plen, lo[q], hi[q] - lo[q], q, (uint32_t)min(total, (uint64_t)0xFFFFFFFFu) };
        }
      atomicAdd(n_hits, (unsigned long long)total);
      nof += total;
      if (nof >= max_mem) break;
    }
    bool ok = false;
    if (!has_hit) {
      const int c = base2(pat[start + plen]);
      if (c >= 0) {
        uint32_t na[PSIGPU_MAX_PARTS], nb[PSIGPU_MAX_PARTS];
#pragma unroll
        for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q) {
          na[q] = nb[q] = 0;
          if (hi[q] > lo[q]) {
            na[q] = mem_lower(mp.p[q].sa, mp.p[q].text4, mp.p[q].n, lo[q], hi[q], plen, c);
            nb[q] = mem_lower(mp.p[q].sa, mp.p[q].text4, mp.p[q].n, na[q], hi[q], plen, c + 1);
            ok = ok || nb[q] > na[q];
          }
        }
        if (ok) {
#pragma unroll
          for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q) { lo[q] = na[q]; hi[q] = nb[q]; }
        }
      }
    }
    if (!ok) {
#pragma unroll
      for (uint32_t q = 0; q < PSIGPU_MAX_PARTS; ++q) { lo[q] = 0; hi[q] = q < mp.n_parts ? mp.p[q].n : 0u; }
      start += (uint64_t)plen + 1; plen = 0; has_hit = false;
      continue;
    }
    ++plen;
  }
}

struct MemHit { uint64_t node_id, node_offset, read_id, read_offset, match_len, gocc; };
static_assert(sizeof(MemHit) == sizeof(psigpu_mem_hit), "MEM record layout");

__global__ void __launch_bounds__(256)
k_mem_locate(const MemGroup* __restrict__ groups, const uint64_t* __restrict__ group_off, uint64_t n_groups,
             MemParts mp, uint64_t rec_offset, MemHit* __restrict__ out)
{
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const MemGroup mg = groups[g];
  const MemPart& pt = mp.p[mg.part];
  MemHit* dst = out + group_off[g];
  for (uint32_t i = 0; i < mg.cnt; ++i) {
    const uint32_t pos = pt.sa[mg.lo + i];
    uint32_t d = pt.seg_dir[pos >> DIR_SHIFT];
    while (pt.seg[d + 1].start <= pos) ++d;
    const SegRec sr = pt.seg[d];
    dst[i] = MemHit{ sr.node_id, (uint64_t)sr.noff + (pos - sr.start), rec_offset + mg.read, mg.start, mg.plen, mg.total };
  }
}

__global__ void k_mem_counts(const MemGroup* __restrict__ groups, uint64_t n, uint32_t* __restrict__ cnt)
{
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) cnt[g] = groups[g].cnt;
}

// ------------------------------------------------------------------------------------
// The host entry's wire format.  A hit record is 4 x u64 (psi::Seed<> as psikt writes it), but of its 32 bytes
// only about 12 carry information: the link out of the device is the bound of the host entry (224 MB of
// records against 158 MB of reads per 1 M-read chunk), so the records cross it as 4 x u32 -- node id minus the
// graph's first id (ids that are rank + constant), node offset, read id minus the sub-batch's first, read offset
// -- and host threads widen them into the caller's 32-byte records while the next sub-batch is in flight.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_hits_wire16(const psigpu_hit* __restrict__ hits, const unsigned long long* __restrict__ n_a, const unsigned long long* __restrict__ n_b,
              uint64_t n_fixed, uint64_t cap, uint64_t id_base, uint64_t rec_base, uint4* __restrict__ out)
{
  // the number of hits: on the device (n_a [+ n_b]) when the host does not know it yet, else n_fixed
  const uint64_t n = min(n_a ? (uint64_t)*n_a + (n_b ? (uint64_t)*n_b : 0ull) : n_fixed, cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
    const ulonglong2 a = src[0], b = src[1];
    out[i] = make_uint4((uint32_t)(a.x - id_base), (uint32_t)a.y, (uint32_t)(b.x - rec_base), (uint32_t)b.y);
  }
}

// Round 4: 8 bytes per record.  The four fields of a hit need far fewer than 64 bits together -- the device sorter
// already packs them into one 64-bit key (hits_gpu.hip) -- so a sub-batch's records cross the link as ONE u64 each:
//     [ read id - the sub-batch's first | read offset | node id - the graph's first id | node offset ]
// with the node fields sized by the graph (bits for its largest node length and its node count), and the read offset
// given every bit the read id of the sub-batch leaves (chr22-like: 17 + 19 + 22 + 6).  The kernel CHECKS that every
// field fits -- the longest read is not known to the host when the kernel is queued -- and raises a flag in mapped host
// memory when one does not; the host entry then makes 16-byte records of that sub-batch instead.  56 MB instead of
// 112 (round 3) or 224 (rounds 1-2) per 1 M-read chunk.
struct WireFmt {
  uint32_t bytes = 0;                          // 8 or 16 (0: no wire records)
  uint32_t noff_bits = 0, node_bits = 0, roff_bits = 0;      // W8; the read id has the remaining 64 - sum bits
};

__global__ void __launch_bounds__(256)
k_hits_wire8(const psigpu_hit* __restrict__ hits, const unsigned long long* __restrict__ n_a, const unsigned long long* __restrict__ n_b,
             uint64_t n_fixed, uint64_t cap, uint64_t id_base, uint64_t rec_base, WireFmt f, uint64_t* __restrict__ out,
             unsigned long long* __restrict__ overflow /* mapped host memory */)
{
  const uint64_t n = min(n_a ? (uint64_t)*n_a + (n_b ? (uint64_t)*n_b : 0ull) : n_fixed, cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t rid_bits = 64 - f.noff_bits - f.node_bits - f.roff_bits;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
    const ulonglong2 a = src[0], b = src[1];
    const uint64_t node = a.x - id_base, rid = b.x - rec_base;
    bad = bad || (a.y >> f.noff_bits) || (node >> f.node_bits) || (b.y >> f.roff_bits) || (rid_bits < 64 && (rid >> rid_bits));
    out[i] = ((((rid << f.roff_bits) | b.y) << f.node_bits | node) << f.noff_bits) | a.y;
  }
  if (__any(bad) && lane_id() == 0) *overflow = 1ull;
}

// ------------------------------------------------------------------------------------
// The part's random-access rate, measured in place (psigpu_measure_random_loads): what the probe of the k-mer
// table (one divergent 16-byte load per lane) and the LF / locate kernels (one 64-byte sector per quad) are
// bounded by.  QUAD = false: every lane loads 16 bytes from a sector of its own; QUAD = true: the four lanes
// of a quad load the four 16-byte pieces of one sector.  Addresses come from a hash of the thread and the
// iteration; `iters` loads per thread, each depending on nothing.
// ------------------------------------------------------------------------------------
template <bool QUAD>
__global__ void __launch_bounds__(256) k_rand_loads(const uint4* __restrict__ t, uint64_t n_sectors, uint32_t iters, uint32_t* out)
{
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x = (QUAD ? (tid >> 2) : tid) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
    const uint64_t sct = __umul64hi(x, n_sectors);
    const uint4 a = t[sct * 4 + (QUAD ? (tid & 3) : ((x >> 5) & 3))];
    acc ^= a.x + a.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// ------------------------------------------------------------------------------------
// Host-side plumbing
// ------------------------------------------------------------------------------------
// PSIGPU_POISON=<byte> (debugging aid): every fresh device allocation is filled with that byte, so that a kernel reading
// what nobody wrote gives the same wrong answer every time instead of whatever the memory held before
static int poison_byte()
{
  static const int b = [] { const char* e = getenv("PSIGPU_POISON"); return e ? (int)(strtoul(e, nullptr, 0) & 0xFF) : -1; }();
  return b;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // uncached: device memory the GPU's L2 does not keep (MTYPE UC); an experiment for the buffers that a copy ENGINE
  // writes or reads behind the runtime's back (PSIGPU_UNCACHED_IO, see psigpu_create)
  bool uncached = false;
  hipError_t ensure(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipErrorUnknown;
    if (uncached) { e = hipExtMallocWithFlags(&p, want, hipDeviceMallocUncached); if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; } }
    if (e != hipSuccess) e = hipMalloc(&p, want);
    if (e == hipSuccess) { cap = want; if (poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), want); (void)hipDeviceSynchronize(); } }
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct TmpBuf {           // scoped device allocation (table construction)
  void* p = nullptr;
  ~TmpBuf() { drop(); }
  void drop() { if (p) (void)hipFree(p); p = nullptr; }
  hipError_t alloc(size_t bytes)
  {
    drop();
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipSuccess && poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), bytes ? bytes : 16); (void)hipDeviceSynchronize(); }
    return e;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// fn(begin, end) over [0, n) on a few host threads (index-load time loops over all nodes / bases)
template <typename F>
void parallel_for(uint64_t n, uint64_t grain, F fn)
{
  unsigned hw = std::thread::hardware_concurrency();
  uint64_t parts = std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 32), (n + grain - 1) / std::max<uint64_t>(1, grain));
  if (parts <= 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + parts - 1) / parts;
  for (uint64_t t = 1; t < parts; ++t) th.emplace_back([=] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
  fn(0, std::min(n, per));
  for (auto& t : th) t.join();
}

}  // namespace

struct psigpu_ctx {
  int device = 0;
  std::string err;
  // graph
  bool have_graph = false;
  uint64_t n_nodes = 0;
  std::vector<uint32_t> node_len;   // label length per node (host): a loaded index's loci are checked against it
  DevBuf nodes, lite, node_id, lab2, labn, edge_to;
  // index
  bool have_index = false;
  bool fm_ok = true;               // rank blocks present: FM search possible (false: k-mer table mode only)
  uint32_t index_k = 0, sa_rate = 0, context = 0, n_paths = 0;
  uint64_t n_loci = 0;
  // One PART of the index on the device: a complete FM index over a group of paths (an index is one part
  // unless its text would pass the 32-bit row limit).  The FM modes and MEM mode search every part; the
  // k-mer table tabulates them together.
  struct FmPart {
    DevBuf blocks, samples, exc_row, exc_sa, ftab, text4, seg, seg_dir, seg_rank;      // (exc_row: + the super-block counts)
    DevBuf saloc;                    // (node rank, offset) per SA row (sa_rate 1, when memory is plentiful)
    DevBuf sarec;                    // per-row records for seed length sarec_k (sa_rate 1, interval table, text resident)
    DevBuf ftabx;                    // interval table with the first row's record in its entries (FtabX), for seed length ftabx_k
    uint64_t text_len = 0, n_exc = 0, n_segs = 0;
    uint64_t C[4] = { 0, 0, 0, 0 };
    uint32_t ftab_len = 0, exc_shift = EXC_SUPER_SHIFT, sarec_k = 0, ftabx_k = 0;
    bool have_text4 = false, have_saloc = false;
    void release()
    {
      for (DevBuf* b : { &blocks, &samples, &exc_row, &exc_sa, &ftab, &text4, &seg, &seg_dir, &seg_rank, &saloc, &sarec, &ftabx }) b->release();
      text_len = n_exc = n_segs = 0; ftab_len = sarec_k = ftabx_k = 0; have_text4 = have_saloc = false;
    }
  };
  std::vector<std::unique_ptr<FmPart>> parts;      // parts[0] always exists
  FmPart& p0() const { return *parts[0]; }
  bool rows_tried = false;         // build_row_records has run for this index (the records exist, or do not fit / apply)
  bool id_affine = false;          // external node id = rank + id_base
  uint64_t id_base = 0;
  DevBuf loci;
  uint32_t gocc_thr = 0;
  uint32_t tune = 0;               // PSIGPU_TUNE_* measurement switches (psigpu_set_tuning)
  bool kt_dedup = false;           // the k-mer table was built without a gocc threshold: one entry per graph position
  // locus k-mer table (built on first use for the index's seed length)
  uint32_t query_mode = PSIGPU_MODE_KMER_TABLE, walk_cap = 0;
  bool lkt_ready = false, lkt_failed = false;
  bool kt_ready = false;           // the table also holds the path k-mers (KmerSlot), K1 is one probe
  DevBuf kt_ht, kt_ext, kt_onpos;
  uint64_t kt_ht_size = 0, kt_n_path_kmers = 0, kt_n_ext = 0;
  uint32_t lkt_k = 0;
  DevBuf lkt_ht, lkt_ent, lkt_res;
  // traverse mode, k > 12: the loci's 12-base prefix walks (k_traverse's pfx_roots), made once per index
  DevBuf pfx_roots;
  uint64_t pfx_n = 0;
  bool pfx_ready = false, pfx_failed = false;
  float pfx_build_ms = 0.f;
  uint64_t lkt_ht_size = 0, lkt_n_ent = 0, lkt_n_res = 0, lkt_n_walks = 0;
  float lkt_build_ms = 0.f;
  std::string lkt_note;
  DevBuf w_seedout, w_seedres, w_iv_tiles_off, w_defer, w_hit_a, w_hit_seed;
  DevBuf w_tilestate;              // k_kmer_step's look-back words, one per tile (a word carries the serial of the call that wrote it)
  const void* tilestate_clean = nullptr;      // the allocation that was last zeroed whole
  bool opt_no_fused = false;       // A/B, tests: the default step as three kernels (k_seed_pack, k_kmer_probe, k_kmer_emit)
  // per-call workspace (grow-only)
  DevBuf in_bases;                 // host entry, reads in pinned memory: the chunk's reads (transfers queued ahead of the compute loop)
  DevBuf in_mask;                  // ... packed reads: their "not ACGT" bits
  DevBuf w_bases, w_read_off, w_cnt, w_tiles, w_seed_off, w_seed_key, w_seed_info,
      w_seed_next, w_ht, w_pfx, w_pfx12, w_iv_lo, w_iv_cnt, w_iv_aux, w_hit_off, w_iv_tiles,
      w_chunks, w_chunk_fill, w_chunk_off, w_chunk_tiles, w_hits, w_spill_a, w_spill_b, w_ctr, w_total,
      w_sb_cnt, w_sb_off, w_sb_tiles, w_sb_key,      // the partition of a chunk's seeds (k_sb_*): counts, offsets, (k-mer, seed) records
      w_seed_wide, w_seed_pfx;                                  // two-word seeds: the k-mers themselves, their first 14 bases
  uint64_t hits_cap_hint = 0, chunks_cap_hint = 0;
  uint64_t spill_cap = 1u << 22;   // traverser spill queue entries (grows when a chunk overflows it)
  void* h_pinned = nullptr;        // pinned host mirror of the counters + counts, written by k_publish
  void* h_pinned_dev = nullptr;    // the same memory as the device addresses it
  hipEvent_t ev[12];
  bool have_events = false;
  hipStream_t stream2 = nullptr;
  psigpu_counters last{};
  uint64_t last_max_read_len = 0;  // longest read of the last run_pipeline call
  // sort-unique on the device (PSIGPU_SORT_UNIQUE)
  uint64_t max_node_len = 0;
  DevBuf ids_sorted;               // node ids in increasing order (only when they are not rank + id_base)
  HitSorter sorter;
  int grouped_state = 0;           // last run_pipeline: 0 groups not looked at, 1 each seed's hits ordered in place and that
                                   // makes the array sorted and duplicate-free, 2 it does not
  DevBuf w_sorted[2], w_count;
  // host entry point: sub-batches of a chunk pipelined through two slots (H2D | kernels | D2H)
  struct Slot {
    DevBuf bases, off, mask;                     // (mask: packed reads, the sub-batch's "not ACGT" bits)
    void* h_stage = nullptr; size_t h_cap = 0;   // pinned staging: rebased read offsets, and the bases of pageable callers
    void* h_stage_dev = nullptr;                 // the same memory as the device addresses it
    hipEvent_t in_ready = nullptr, out_done = nullptr;
    DevBuf d_wire;                               // 16-byte wire records of the slot's sub-batch (k_hits_wire16)
    void* h_wire = nullptr; size_t h_wire_cap = 0;   // pinned: where they land on the host, before they are widened
  } slot[2];
  DevBuf w_hits_alt;
  hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
  // the pipeline's transfers, each direction on a copy engine of its own (see pipeline_init)
  struct EngineCopy {
    bool ok = false;
    int n_sig = 0;                               // signals taken from the process-wide pool: sig_in[0..IN_RING-1], then sig_out[0..1]
    static constexpr int IN_RING = 8;            // transfers of reads that may be queued ahead of the compute loop
    hsa_agent_t gpu{}, cpu{};
    uint32_t eng_in = 0, eng_out = 0;            // hsa_amd_sdma_engine_id_t bits
    hsa_signal_t sig_in[IN_RING]{}, sig_out[2]{};      // 1 while the transfer is in flight (two-slot path: sig_in[0..1])
    hsa_signal_t sig_fast[3]{};                        // ... the transfers out of the lookahead path's three slots
  } ec;
  double hits_per_read_hint = 0.0;
  bool trace_call = false;         // the host-entry call in progress runs under PSIGPU_TRACE
  // host entry, default mode: two sub-batches in flight (the kernels of sub-batch i + 1 are queued before the host waits
  // for sub-batch i).  Set by a run_pipeline call that went through the default mode's five kernels alone; per in-flight
  // sub-batch a hit buffer, a mapped block for its counters and an event.
  uint32_t fast_k = 0, fast_flags = 0;
  bool fast_on = false, fast_off = false;
  static constexpr int N_FAST = 3;     // (two in the queue + the one whose records are on their way out)
  struct FastSlot {
    DevBuf hits, off, wire;
    void* h = nullptr; void* h_dev = nullptr;
    void* h_wire = nullptr; size_t h_wire_cap = 0;
    hipEvent_t begin = nullptr, done = nullptr;
  } fast[N_FAST];
  bool opt_no_lookahead = false;
  uint64_t lookahead_fallbacks = 0;
  // what the graph and the index left on the device, with a checksum of every array taken when it was loaded
  // (psigpu_verify_resident: has anything of it changed since?)
  struct Resident { std::string name; const DevBuf* buf; const void* at; uint64_t bytes; uint64_t sum; };
  std::vector<Resident> resident;
  // device-resident entry, two chunks in flight (psigpu_find_seeds_device_begin / _end): what was begun and not ended yet,
  // oldest first.  A chunk that could be queued (the default mode's kernels alone: enqueue_default) sits in a FastSlot;
  // any other chunk is answered by the synchronous entry when its turn to be ended comes.
  struct DevPending {
    bool queued = false;
    const char* d_bases = nullptr; const uint64_t* d_mask = nullptr; bool packed = false; const uint64_t* d_off = nullptr;
    uint64_t nr = 0, nb = 0, rec_offset = 0; uint32_t k = 0, step = 0, flags = 0; void* stream = nullptr;
    unsigned long long serial = 0; bool uniform = false; uint64_t cap = 0; int slot = 0;
  } dpend[2];
  int dp_head = 0, dp_count = 0;
  uint64_t dp_seq = 0;
  void* dp_stream = nullptr;       // the stream of the chunks in flight (one stream for all of them: the workspace is shared)
  std::vector<void*> retired;      // hit buffers outgrown by a _begin while a caller may still read them: freed by the next _end
  void* stager = nullptr;          // the host entry's helper thread for pageable reads (struct Worker)
  void* widener = nullptr;         // the host entry's widening threads (struct Widener, made by its first call)
  // psigpu_set_option
  uint64_t opt_sub_bytes = 0;
  bool opt_no_ahead = false, opt_no_engine_copy = false;
  uint32_t opt_wire = 0;           // 0: the narrowest wire record that fits; 8 / 16 / 32: nothing narrower
  uint32_t opt_wire8_roff_cap = 0; // test hook: at most this many read-offset bits in an 8-byte record
  bool opt_no_pfx_roots = false;   // traverse mode from the loci themselves (A/B, tests)
  bool opt_res16 = false;          // 16 bytes of probe results per seed (A/B, tests)
  uint64_t opt_expected_calls = 0; // PSIGPU_MODE_AUTO: chunks the caller expects to ask (0: unknown)
  uint64_t opt_expected_seeds = 0; // ... and seeds over all of them
  bool auto_mode = false, auto_resolved = false;
  uint32_t wire_used = 0;          // bytes per wire record the last run_pipeline call left in its wire buffer (0: none)
  unsigned long long serial = 0;   // run_pipeline calls so far: every call's counter block carries its number
  uint64_t uniform_refuted = 0;    // calls that claimed PSIGPU_UNIFORM_READS for reads that were not (answered again the general way)
  uint64_t stale_handbacks = 0;    // counter blocks that came back with another call's number (psigpu_counters.stale_handbacks)
  bool wire8_overflowed = false;   // a sub-batch's records did not fit 8 bytes: the context stays with 16 from then on
};

static uint32_t bits_for(uint64_t max_value)      // bits needed to hold 0..max_value (at least 1)
{
  uint32_t b = 1;
  while (b < 64 && (max_value >> b)) ++b;
  return b;
}

static thread_local std::string g_create_err;

// The host entry's HSA objects live as long as the process: one reference on the runtime (HIP holds its own), and the
// completion signals of the engine copies are handed from context to context instead of being destroyed with one --
// ROCr may still be retiring a copy on its own thread when the waiter that saw the signal reach 0 is already
// tearing the context down (a finder closed right after its last chunk, under load: silent SIGSEGVs and
// "double free or corruption" in one of every ~8 fuzz processes sharing a box, none since).
namespace {
struct HsaGlobals {
  std::mutex mu;
  bool tried = false, up = false;
  std::vector<hsa_signal_t> idle;
  bool init()
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!tried) { tried = true; up = hsa_init() == HSA_STATUS_SUCCESS; }
    return up;
  }
  bool take(hsa_signal_t* sg)
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!idle.empty()) { *sg = idle.back(); idle.pop_back(); hsa_signal_store_relaxed(*sg, 0); return true; }
    }
    return hsa_signal_create(0, 0, nullptr, sg) == HSA_STATUS_SUCCESS;
  }
  void give(hsa_signal_t sg) { std::lock_guard<std::mutex> lk(mu); idle.push_back(sg); }
};
HsaGlobals g_hsa;
}  // namespace

// Pinned host buffers for returned hits are recycled process-wide: hipHostMalloc of a few
// hundred MB costs tens of milliseconds, a chunk loop would pay it every call.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> free_list, live;
  void* get(size_t bytes)
  {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].second >= bytes) {
        auto b = free_list[i];
        free_list.erase(free_list.begin() + i);
        live.push_back(b);
        return b.first;
      }
    void* p = nullptr;
    size_t want = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocMapped) != hipSuccess) return nullptr;
    live.emplace_back(p, want);
    return p;
  }
  void put(void* p)
  {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < live.size(); ++i)
      if (live[i].first == p) {
        free_list.push_back(live[i]);
        live.erase(live.begin() + i);
        // keep at most two idle buffers
        while (free_list.size() > 2) { (void)hipHostFree(free_list.front().first); free_list.erase(free_list.begin()); }
        return;
      }
    (void)hipHostFree(p);
  }
};
PinnedPool g_pinned;
}  // namespace

#define HIPCHK(ctx, call)                                                                    \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
      return PSIGPU_ERR_DEVICE;                                                              \
    }                                                                                        \
  } while (0)

// A large array of ordinary (pageable) host memory to the device: hipMemcpy stages such a copy on one thread
// (6-10 GB/s: a whole-genome index, 45 GB, took 7 of the 11 s of psigpu_load_index); here a few threads copy
// 32-MiB pieces into two pinned buffers while the previous piece is on its way.
static void copy_on_threads(char* dst, const char* src, size_t n)
{
  const unsigned hw = std::thread::hardware_concurrency();
  const unsigned parts = (unsigned)std::min<size_t>(std::min<unsigned>(8, hw ? hw : 1), std::max<size_t>(1, n / (2u << 20)));
  if (parts <= 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n / parts + 63) & ~(size_t)63;
  for (unsigned t = 1; t < parts; ++t) {
    const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per);
    th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
  }
  memcpy(dst, src, std::min(n, per));
  for (auto& t : th) t.join();
}

static int upload_large(psigpu_ctx* ctx, void* dst, const void* src, size_t bytes)
{
  constexpr size_t PIECE = 32u << 20;
  struct Stage {
    void* buf[2] = { nullptr, nullptr };
    hipEvent_t done[2] = { nullptr, nullptr };
    hipStream_t s = nullptr;
    ~Stage()
    {
      if (s) (void)hipStreamSynchronize(s);       // (an error path may leave a piece in flight)
      for (int i = 0; i < 2; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (done[i]) (void)hipEventDestroy(done[i]); }
      if (s) (void)hipStreamDestroy(s);
    }
  } st;
  bool ok = hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipHostMalloc(&st.buf[i], PIECE, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&st.done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {                                        // no pinned memory to spare: the plain copy
    (void)hipGetLastError();
    HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PSIGPU_OK;
  }
  size_t piece = 0;
  for (size_t off = 0; off < bytes; off += PIECE, ++piece) {
    const size_t len = std::min(PIECE, bytes - off);
    const int i = (int)(piece & 1);
    if (piece >= 2) HIPCHK(ctx, hipEventSynchronize(st.done[i]));
    copy_on_threads((char*)st.buf[i], (const char*)src + off, len);
    HIPCHK(ctx, hipMemcpyAsync((char*)dst + off, st.buf[i], len, hipMemcpyHostToDevice, st.s));
    HIPCHK(ctx, hipEventRecord(st.done[i], st.s));
  }
  HIPCHK(ctx, hipStreamSynchronize(st.s));
  return PSIGPU_OK;
}

// A/B switches of the load campaigns (read per call: a campaign sets them for its own finders).
// PSIGPU_AB_LOAD_HOLE=1 brings back what the loaders did before round 5: pads filled on the null stream with nobody waiting,
// no device synchronisation when a loader returns, no read-back of checksums (which happened to order the fills).
// PSIGPU_AB_NO_PAD_ZERO=1 leaves the pads as allocated (with PSIGPU_POISON: a known byte) -- does any answer depend on them?
static bool ab_load_hole() { return getenv("PSIGPU_AB_LOAD_HOLE") != nullptr; }
static bool ab_no_pad_zero() { return getenv("PSIGPU_AB_NO_PAD_ZERO") != nullptr; }
// every loader ends here: whatever it queued on any stream (fills, table kernels) has run when the caller gets control back
static int loader_fence(psigpu_ctx* ctx)
{
  if (ab_load_hole()) return PSIGPU_OK;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { ctx->err = std::string("hipDeviceSynchronize (end of a loader): ") + hipGetErrorString(e); return PSIGPU_ERR_DEVICE; }
  return PSIGPU_OK;
}

// ---- what the device holds of the graph and the index, checked against what was put there ------------------------
// Three wrong answers of the load campaigns have in common the data a freshly loaded finder reads, not a
// kernel.  Every array the loaders put on the device leaves a 64-bit checksum behind (a grid-stride kernel: 9 GB in a few
// milliseconds); psigpu_verify_resident recomputes them -- "has anything the finder reads changed since it was loaded?" --
// and with PSIGPU_VERIFY_UPLOAD=1 every upload is also checked against the same sum over its HOST source (one pass of
// the CPU over the array: campaigns only).
__device__ __host__ inline uint64_t resident_mix(uint64_t w, uint64_t i)
{
  uint64_t x = w + 0x9E3779B97F4A7C15ull * (i + 1);
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
  return x;
}
__global__ void __launch_bounds__(256) k_checksum(const uint64_t* __restrict__ p, uint64_t n_words, unsigned long long* __restrict__ out)
{
  uint64_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) acc += resident_mix(p[i], i);
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
  if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}
static int device_checksum(psigpu_ctx* ctx, const void* d, uint64_t bytes, uint64_t* sum)
{
  *sum = 0;
  const uint64_t n_words = bytes / 8;                 // (a tail of fewer than eight bytes is left out on both sides)
  if (n_words == 0) return PSIGPU_OK;
  TmpBuf acc;
  HIPCHK(ctx, acc.alloc(8));
  HIPCHK(ctx, hipMemset(acc.p, 0, 8));
  const unsigned grid = (unsigned)std::min<uint64_t>((n_words + 255) / 256, 256 * 16);
  k_checksum<<<grid, 256>>>(reinterpret_cast<const uint64_t*>(d), n_words, acc.as<unsigned long long>());
  HIPCHK(ctx, hipMemcpy(sum, acc.p, 8, hipMemcpyDeviceToHost));
  return PSIGPU_OK;
}
static uint64_t host_checksum(const void* h, uint64_t bytes)
{
  const uint64_t n_words = bytes / 8;
  std::atomic<uint64_t> acc{ 0 };
  const char* c = static_cast<const char*>(h);
  parallel_for(n_words, 1u << 20, [&](uint64_t i0, uint64_t i1) {
    uint64_t a = 0;
    for (uint64_t i = i0; i < i1; ++i) { uint64_t w; memcpy(&w, c + 8 * i, 8); a += resident_mix(w, i); }
    acc.fetch_add(a, std::memory_order_relaxed);
  });
  return acc.load();
}
// record (and, on request, verify against the host source) what was just put at `at` (inside b)
static int resident_note(psigpu_ctx* ctx, const char* name, const DevBuf& b, const void* at, const void* host_src, uint64_t bytes)
{
  const bool env_verify = getenv("PSIGPU_VERIFY_UPLOAD") != nullptr;      // (read per load: a campaign switches it on for its own finders)
  uint64_t sum = 0;
  int st = device_checksum(ctx, at, bytes, &sum);
  if (st != PSIGPU_OK) return st;
  if (env_verify && host_src && sum != host_checksum(host_src, bytes)) {
    fprintf(stderr, "[psigpu] PSIGPU_VERIFY_UPLOAD: %s (%llu bytes) is not on the device what it is on the host\n", name, (unsigned long long)bytes);
    ctx->err = std::string("upload of ") + name + " failed verification";
    return PSIGPU_ERR_DEVICE;
  }
  for (auto& r : ctx->resident)
    if (r.buf == &b && r.at == at) { r.name = name; r.bytes = bytes; r.sum = sum; return PSIGPU_OK; }
  ctx->resident.push_back(psigpu_ctx::Resident{ name, &b, at, bytes, sum });
  return PSIGPU_OK;
}
static void resident_forget(psigpu_ctx* ctx, const DevBuf& b)
{
  auto& v = ctx->resident;
  v.erase(std::remove_if(v.begin(), v.end(), [&](const psigpu_ctx::Resident& r) { return r.buf == &b; }), v.end());
}

template <typename T>
static int upload(psigpu_ctx* ctx, DevBuf& b, const T* src, uint64_t n, uint64_t pad_elems = 0, const char* name = nullptr)
{
  if (name) resident_forget(ctx, b);
  HIPCHK(ctx, b.ensure((n + pad_elems) * sizeof(T) + 16));
  if (n * sizeof(T) >= (64u << 20)) { int st = upload_large(ctx, b.p, src, n * sizeof(T)); if (st != PSIGPU_OK) return st; }
  else
  if (n) HIPCHK(ctx, hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  if (pad_elems && !ab_no_pad_zero()) {
    // hipMemset on the null stream returns before the fill has run, and the query kernels run on non-blocking streams that
    // do not order against the null stream: the host waits for the fill here (round-4 review: the loaders' ordering hole)
    HIPCHK(ctx, hipMemsetAsync((char*)b.p + n * sizeof(T), 0, pad_elems * sizeof(T), nullptr));
    if (!ab_load_hole()) HIPCHK(ctx, hipStreamSynchronize(nullptr));
  }
  if (name && n && !ab_load_hole()) return resident_note(ctx, name, b, b.p, src, n * sizeof(T));
  return PSIGPU_OK;
}

extern "C" {

// PSIGPU_SEGV_TRACE=1 (debugging aid): the signal, the faulting address, the instruction pointer and the thread on stderr FIRST
// (formatted by hand: nothing here may allocate), then a backtrace of the faulting thread -- which walks the stack that may be
// the thing that is broken; a process that died inside it used to leave nothing behind.
static void segv_put_hex(char*& p, unsigned long long v)
{
  *p++ = '0'; *p++ = 'x';
  bool on = false;
  for (int sh = 60; sh >= 0; sh -= 4) {
    const unsigned d = (unsigned)((v >> sh) & 15u);
    if (d || on || sh == 0) { *p++ = (char)(d < 10 ? '0' + d : 'a' + d - 10); on = true; }
  }
}
static void segv_trace(int sig, siginfo_t* si, void* uc_)
{
  signal(sig, SIG_DFL);
  alarm(5);                                       // (a handler stuck behind a lock the dying thread holds must not hang the process)
  char line[256];
  char* p = line;
  for (const char* c = "[psigpu] fatal signal "; *c; ++c) *p++ = *c;
  segv_put_hex(p, (unsigned long long)sig);
  for (const char* c = " address "; *c; ++c) *p++ = *c;
  segv_put_hex(p, (unsigned long long)(uintptr_t)(si ? si->si_addr : nullptr));
#if defined(__x86_64__)
  if (uc_) {
    const ucontext_t* uc = static_cast<const ucontext_t*>(uc_);
    for (const char* c = " rip "; *c; ++c) *p++ = *c;
    segv_put_hex(p, (unsigned long long)uc->uc_mcontext.gregs[REG_RIP]);
    for (const char* c = " rsp "; *c; ++c) *p++ = *c;
    segv_put_hex(p, (unsigned long long)uc->uc_mcontext.gregs[REG_RSP]);
  }
#endif
  for (const char* c = " thread "; *c; ++c) *p++ = *c;
  segv_put_hex(p, (unsigned long long)syscall(SYS_gettid));
  for (const char* c = "; backtrace of the faulting thread:\n"; *c; ++c) *p++ = *c;
  (void)!write(2, line, (size_t)(p - line));
  void* frames[64];
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  raise(sig);
}

psigpu_ctx* psigpu_create(int device)
{
  static const bool traced = [] {
    if (!getenv("PSIGPU_SEGV_TRACE")) return false;
    static char alt[1 << 16];
    { void* warm[4]; (void)backtrace(warm, 4); }      // (the first call loads libgcc: not from inside a handler)
    stack_t ss{}; ss.ss_sp = alt; ss.ss_size = sizeof alt;
    (void)sigaltstack(&ss, nullptr);
    struct sigaction sa{};
    sa.sa_sigaction = segv_trace; sa.sa_flags = SA_ONSTACK | SA_SIGINFO;
    for (int sg : { SIGSEGV, SIGBUS, SIGABRT, SIGFPE }) (void)sigaction(sg, &sa, nullptr);
    return true;
  }();
  (void)traced;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_err = "no HIP device available (this library has no CPU fallback)";
    return nullptr;
  }
  if (device < 0 || device >= n) { g_create_err = "device ordinal out of range"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return nullptr; }
  psigpu_ctx* ctx = new psigpu_ctx;
  ctx->device = device;
  if (getenv("PSIGPU_UNCACHED_IO")) {           // experiment: the buffers the copy engines write / read as MTYPE UC memory
    ctx->in_bases.uncached = true; ctx->in_mask.uncached = true;      // (tried as the default against rare wrong records under load: a test that
    for (auto& sl : ctx->slot) { sl.bases.uncached = true; sl.mask.uncached = true; sl.d_wire.uncached = true; }      // had never failed failed 3 times in 24 -- not kept)
  }
  ctx->parts.emplace_back(new psigpu_ctx::FmPart);
  for (auto& ev : ctx->ev)
    if (hipEventCreate(&ev) != hipSuccess) { g_create_err = "hipEventCreate failed"; delete ctx; return nullptr; }
  ctx->have_events = true;
  if (hipHostMalloc(&ctx->h_pinned, sizeof(DevCounters) + 64, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer(&ctx->h_pinned_dev, ctx->h_pinned, 0) != hipSuccess) {
    g_create_err = "hipHostMalloc failed"; psigpu_destroy(ctx); return nullptr;
  }
  return ctx;
}

static void widener_destroy(psigpu_ctx* ctx);

void psigpu_destroy(psigpu_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // nothing of this context may still be running or landing when its memory goes back to the allocator: every stream
  // (chunks begun and never ended, a caller's stream the last call ran on), then every engine copy -- those are not
  // HIP's and no HIP call waits for them (bounded: a signal nobody will ever lower must not hang a destructor)
  (void)hipDeviceSynchronize();
  ctx->dp_count = 0;
  for (int i = 0; i < ctx->ec.n_sig; ++i) {
    constexpr int R = psigpu_ctx::EngineCopy::IN_RING;
    const hsa_signal_t sg = i < R ? ctx->ec.sig_in[i] : i < R + 2 ? ctx->ec.sig_out[i - R] : ctx->ec.sig_fast[i - R - 2];
    for (int spin = 0; spin < 2000 && hsa_signal_wait_scacquire(sg, HSA_SIGNAL_CONDITION_LT, 1, 1000000, HSA_WAIT_STATE_BLOCKED) >= 1; ++spin) { }
  }
  for (void* old : ctx->retired) (void)hipFree(old);
  ctx->retired.clear();
  widener_destroy(ctx);
  DevBuf* all[] = { &ctx->nodes, &ctx->lite, &ctx->node_id, &ctx->lab2, &ctx->labn, &ctx->edge_to, &ctx->loci, &ctx->w_bases,
                    &ctx->w_read_off, &ctx->w_cnt, &ctx->w_tiles, &ctx->w_seed_off, &ctx->w_seed_key,
                    &ctx->w_seed_info, &ctx->w_seed_next, &ctx->w_ht, &ctx->w_pfx, &ctx->w_pfx12, &ctx->w_iv_lo, &ctx->w_iv_cnt, &ctx->w_iv_aux, &ctx->w_hit_off,
                    &ctx->w_iv_tiles, &ctx->w_chunks, &ctx->w_chunk_fill, &ctx->w_chunk_off, &ctx->w_chunk_tiles, &ctx->w_hits, &ctx->w_spill_a, &ctx->w_spill_b,
                    &ctx->w_ctr, &ctx->w_total, &ctx->lkt_ht, &ctx->lkt_ent, &ctx->lkt_res, &ctx->w_seedout,
                    &ctx->w_iv_tiles_off, &ctx->w_defer, &ctx->kt_ht, &ctx->kt_ext, &ctx->w_seedres };
  for (auto* b : all) b->release();
  ctx->ids_sorted.release(); ctx->w_sorted[0].release(); ctx->w_sorted[1].release(); ctx->w_count.release();
  ctx->kt_onpos.release(); ctx->w_hit_a.release(); ctx->w_hit_seed.release(); ctx->pfx_roots.release(); ctx->w_tilestate.release();
  for (DevBuf* b : { &ctx->w_sb_cnt, &ctx->w_sb_off, &ctx->w_sb_tiles, &ctx->w_sb_key, &ctx->w_seed_wide, &ctx->w_seed_pfx }) b->release();
  for (auto& m : ctx->parts) m->release();
  ctx->w_hits_alt.release(); ctx->in_bases.release(); ctx->in_mask.release();
  for (auto& sl : ctx->slot) {
    sl.bases.release(); sl.off.release(); sl.mask.release(); sl.d_wire.release();
    if (sl.h_stage) (void)hipHostFree(sl.h_stage);
    if (sl.h_wire) (void)hipHostFree(sl.h_wire);
    if (sl.in_ready) (void)hipEventDestroy(sl.in_ready);
    if (sl.out_done) (void)hipEventDestroy(sl.out_done);
  }
  for (auto& fs : ctx->fast) {
    fs.hits.release(); fs.off.release(); fs.wire.release();
    if (fs.h) (void)hipHostFree(fs.h);
    if (fs.h_wire) (void)hipHostFree(fs.h_wire);
    if (fs.begin) (void)hipEventDestroy(fs.begin);
    if (fs.done) (void)hipEventDestroy(fs.done);
  }
  for (hipStream_t st : { ctx->s_in, ctx->s_comp, ctx->s_out }) if (st) (void)hipStreamDestroy(st);
  for (int i = 0; i < ctx->ec.n_sig; ++i) {    // (kept for the next context)
    constexpr int R = psigpu_ctx::EngineCopy::IN_RING;
    g_hsa.give(i < R ? ctx->ec.sig_in[i] : i < R + 2 ? ctx->ec.sig_out[i - R] : ctx->ec.sig_fast[i - R - 2]);
  }
  if (ctx->have_events) for (auto& ev : ctx->ev) (void)hipEventDestroy(ev);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  delete ctx;
}

const char* psigpu_last_error(const psigpu_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

static void lkt_release(psigpu_ctx* ctx);
static void pfx_release(psigpu_ctx* ctx);
static void drop_row_records(psigpu_ctx* ctx);

int psigpu_set_gocc_threshold(psigpu_ctx* ctx, uint32_t thr)
{
  if (!ctx) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  if (thr && ctx->kt_ready && ctx->kt_dedup) {       // the table holds positions, the threshold counts occurrences: rebuilt
    if (hipSetDevice(ctx->device) != hipSuccess) return PSIGPU_ERR_DEVICE;
    lkt_release(ctx);
  }
  ctx->gocc_thr = thr;
  return PSIGPU_OK;
}

static void pfx_release(psigpu_ctx* ctx)
{
  ctx->pfx_roots.release();
  ctx->pfx_n = 0; ctx->pfx_ready = ctx->pfx_failed = false;
}

static void lkt_release(psigpu_ctx* ctx)
{
  ctx->lkt_ht.release(); ctx->lkt_ent.release(); ctx->lkt_res.release(); ctx->kt_ht.release(); ctx->kt_ext.release();
  ctx->kt_onpos.release();
  ctx->lkt_ready = false; ctx->lkt_failed = false; ctx->kt_ready = false;
  ctx->fast_k = 0;
  ctx->kt_ht_size = ctx->kt_n_path_kmers = 0;
  ctx->lkt_ht_size = ctx->lkt_n_ent = ctx->lkt_n_res = ctx->lkt_n_walks = 0;
  ctx->lkt_note.clear();
}

int psigpu_set_query_mode(psigpu_ctx* ctx, uint32_t mode, uint32_t walk_cap)
{
  if (!ctx || mode > PSIGPU_MODE_AUTO) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  if (hipSetDevice(ctx->device) != hipSuccess) return PSIGPU_ERR_DEVICE;
  ctx->auto_mode = mode == PSIGPU_MODE_AUTO;
  ctx->auto_resolved = false;
  if (ctx->auto_mode) { ctx->walk_cap = walk_cap; return PSIGPU_OK; }      // (resolved, and the tables dropped if need be, by resolve_auto_mode)
  if (mode != ctx->query_mode || walk_cap != ctx->walk_cap) lkt_release(ctx);
  ctx->query_mode = mode;
  ctx->walk_cap = walk_cap;
  return PSIGPU_OK;
}

uint32_t psigpu_query_mode(const psigpu_ctx* ctx)
{
  if (!ctx) return PSIGPU_MODE_KMER_TABLE;
  return (ctx->auto_mode && !ctx->auto_resolved) ? PSIGPU_MODE_AUTO : ctx->query_mode;
}

// PSIGPU_MODE_AUTO: k-mer table or traverser, by device time over the expected calls.  The constants are this part's,
// measured on the chr22-like and the whole-genome workloads: the tables cost ~1 ns per tabulated
// k-mer (k-walks of the loci, ~2.6 per locus on SNV graphs, + path positions); without them every call pays the traverser's
// pass over the loci (~40 ps per locus) and the chunk's seed table with its lookups (~50 ps per seed: "expected_seeds").
static void resolve_auto_mode(psigpu_ctx* ctx)
{
  if (!ctx->auto_mode || ctx->auto_resolved || !ctx->have_index) return;
  uint64_t text = 0;
  for (const auto& fp : ctx->parts) text += fp->text_len;
  const double t_build_ms = (2.6 * (double)ctx->n_loci + (double)text) * 1.0e-6;
  const double t_call_ms = (double)ctx->n_loci * 40e-9;
  const double t_seeds_ms = (double)ctx->opt_expected_seeds * 50e-9;         // the chunk's seed table + its lookups, per seed
  const bool table = ctx->opt_expected_calls == 0 || (double)ctx->opt_expected_calls * t_call_ms + t_seeds_ms > t_build_ms;
  const uint32_t mode = table ? PSIGPU_MODE_KMER_TABLE : PSIGPU_MODE_TRAVERSE;
  if (mode != ctx->query_mode) lkt_release(ctx);
  ctx->query_mode = mode;
  ctx->auto_resolved = true;
  if (getenv("PSIGPU_TRACE"))
    fprintf(stderr, "[psigpu] auto mode: %s (tables ~%.1f ms, traverser ~%.2f ms per call, %llu calls expected)\n",
            table ? "k-mer table" : "traverse", t_build_ms, t_call_ms, (unsigned long long)ctx->opt_expected_calls);
}

int psigpu_set_tuning(psigpu_ctx* ctx, uint32_t flags)
{
  if (!ctx) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  if (hipSetDevice(ctx->device) != hipSuccess) return PSIGPU_ERR_DEVICE;
  if ((flags ^ ctx->tune) & PSIGPU_TUNE_NO_ROWRECS) drop_row_records(ctx);      // made (or not) by the next FM query
  if (((flags ^ ctx->tune) & PSIGPU_TUNE_NO_PATH_TABLE) && ctx->query_mode == PSIGPU_MODE_TRAVERSE) lkt_release(ctx);
  ctx->tune = flags;
  return PSIGPU_OK;
}

int psigpu_set_option(psigpu_ctx* ctx, const char* name, uint64_t value)
{
  if (!ctx || !name) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  const std::string n(name);
  if (n == "sub_bytes") ctx->opt_sub_bytes = value;
  else if (n == "no_ahead") ctx->opt_no_ahead = value != 0;
  else if (n == "no_engine_copy") ctx->opt_no_engine_copy = value != 0;
  else if (n == "wire") {
    if (value != 0 && value != 8 && value != 16 && value != 32) { ctx->err = "wire: 0, 8, 16 or 32"; return PSIGPU_ERR_ARG; }
    ctx->opt_wire = (uint32_t)value;
  } else if (n == "no_pfx_roots") ctx->opt_no_pfx_roots = value != 0;
  else if (n == "res16") ctx->opt_res16 = value != 0;
  else if (n == "no_lookahead") ctx->opt_no_lookahead = value != 0;
  else if (n == "no_fused") ctx->opt_no_fused = value != 0;
  else if (n == "expected_calls") { ctx->opt_expected_calls = value; ctx->auto_resolved = false; }
  else if (n == "expected_seeds") { ctx->opt_expected_seeds = value; ctx->auto_resolved = false; }
  else if (n == "corrupt_resident") {                       // (test hook: one word of the starting loci changed on the device)
    if (ctx->loci.p && ctx->n_loci) { const uint32_t v = (uint32_t)value; (void)hipMemcpy(ctx->loci.p, &v, 4, hipMemcpyHostToDevice); }
  }
  else if (n == "wire8_roff_bits") { ctx->opt_wire8_roff_cap = (uint32_t)value; ctx->wire8_overflowed = false; }      // (test hook)
  else { ctx->err = "unknown option '" + n + "'"; return PSIGPU_ERR_ARG; }
  return PSIGPU_OK;
}

int psigpu_load_graph(psigpu_ctx* ctx, const psigpu_graph_view* g)
{
  if (!ctx || !g) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t n = g->n_nodes;
  if (n >= 0xFFFFFFF0ull) { ctx->err = "too many nodes"; return PSIGPU_ERR_ARG; }
  uint64_t total = n ? g->label_off[n] : 0;
  if (total >= (1ull << 40)) { ctx->err = "labels too long"; return PSIGPU_ERR_ARG; }
  std::vector<NodeRec> recs(n);
  std::vector<uint64_t> lab2(total / 32 + 2, 0), labn(total / 64 + 2, 0);
  // 2-bit labels and the N mask: threads take base ranges aligned to 64 bases, so that no two of them
  // touch one word (300 M nodes / 3 G bases at whole-genome size: this loop and the next are the bulk
  // of psigpu_load_graph)
  parallel_for((total + 63) / 64, 1u << 14, [&](uint64_t w0, uint64_t w1) {
    const uint64_t i1 = std::min<uint64_t>(total, w1 * 64);
    for (uint64_t i = w0 * 64; i < i1; ++i) {
      uint64_t two;
      switch (g->labels[i]) {
        case 'A': case 'a': two = 0; break;
        case 'C': case 'c': two = 1; break;
        case 'G': case 'g': two = 2; break;
        case 'T': case 't': two = 3; break;
        default: two = 0; labn[i >> 6] |= 1ull << (63 - (i & 63)); break;
      }
      lab2[i >> 5] |= two << (62 - 2 * (i & 31));
    }
  });
  std::atomic<int> bad{ 0 };
  parallel_for(n, 1u << 14, [&](uint64_t v0, uint64_t v1) {
    for (uint64_t v = v0; v < v1; ++v) {
      const uint64_t o0 = g->label_off[v], o1 = g->label_off[v + 1];
      const uint64_t deg = g->edge_off[v + 1] - g->edge_off[v];
      if (deg > 0xFFFF) { bad = 1; return; }
      if (o1 - o0 > 0xFFFFFFFFull || g->edge_off[v] > 0xFFFFFFFFull) { bad = 2; return; }
      bool has_n = false;
      for (uint64_t w = o0 >> 6; w <= (o1 ? (o1 - 1) >> 6 : 0) && o1 > o0 && !has_n; ++w) {
        uint64_t m = labn[w];
        if (w == (o0 >> 6)) m &= ~0ull >> (o0 & 63);
        if (w == ((o1 - 1) >> 6)) m &= ~0ull << (63 - ((o1 - 1) & 63));
        has_n = m != 0;
      }
      recs[v].w0 = o0 | (deg << 40) | ((uint64_t)has_n << 63);
      recs[v].len = (uint32_t)(o1 - o0);
    }
  });
  if (bad == 1) { ctx->err = "out-degree above 65535"; return PSIGPU_ERR_ARG; }
  if (bad == 2) { ctx->err = "graph too large"; return PSIGPU_ERR_ARG; }
  auto base_at = [&](uint64_t i) { return (lab2[i >> 5] >> (62 - 2 * (i & 31))) & 3ull; };
  auto n_at = [&](uint64_t i) { return (labn[i >> 6] >> (63 - (i & 63))) & 1ull; };
  parallel_for(n, 1u << 14, [&](uint64_t v_begin, uint64_t v_end) {
  for (uint64_t v = v_begin; v < v_end; ++v) {
    uint64_t o0 = g->label_off[v], len = g->label_off[v + 1] - o0;
    uint64_t head2 = 0; uint32_t headn = 0;
    uint64_t tot = 0;                       // inline bases so far
    auto append = [&](uint64_t from, uint64_t cnt) {
      for (uint64_t i = 0; i < cnt; ++i, ++tot) {
        head2 |= base_at(from + i) << (62 - 2 * tot);
        headn |= (uint32_t)n_at(from + i) << (31 - tot);
      }
    };
    NodeRec& r = recs[v];
    uint64_t cur = v, coff = 0;
    bool cut = false;                       // the 32-base cap ended inside a node
    if (len > 32) {
      append(o0, 32);
      r.w0 |= 1ull << 62;
    } else {
      append(o0, len);
      for (int guard = 0; guard < 64 && tot < 32; ++guard) {
        if (g->edge_off[cur + 1] - g->edge_off[cur] != 1) break;
        uint64_t nxt = g->edge_to[g->edge_off[cur]];
        uint64_t nlen = g->label_off[nxt + 1] - g->label_off[nxt];
        uint64_t take = std::min<uint64_t>(nlen, 32 - tot);
        append(g->label_off[nxt], take);
        if (take < nlen) { cur = nxt; coff = take; cut = true; break; }
        cur = nxt;
      }
      r.len = (uint32_t)tot;
    }
    uint64_t deg = cut ? 1 : g->edge_off[cur + 1] - g->edge_off[cur];
    r.w0 = (r.w0 & ~(0xFFFFull << 40)) | (deg << 40) | (coff << 56);
    r.head2 = head2;
    r.headn = headn;
    r.edge0 = cut ? (uint32_t)cur : (deg ? g->edge_to[g->edge_off[cur]] : NIL);
    r.edge_off = cut ? 0u : (deg == 2 ? g->edge_to[g->edge_off[cur] + 1] : (uint32_t)g->edge_off[cur]);
  }
  });
  // compact 16-byte records for the common case
  std::vector<NodeLite> lite(n);
  parallel_for(n, 1u << 14, [&](uint64_t v_begin, uint64_t v_end) {
  for (uint64_t v = v_begin; v < v_end; ++v) {
    const NodeRec& r = recs[v];
    uint64_t deg = (r.w0 >> 40) & 0xFFFF;
    bool is_long = (r.w0 >> 62) & 1;
    int64_t d1 = deg == 2 ? (int64_t)r.edge_off - (int64_t)r.edge0 : 0;
    bool slow = is_long || r.headn != 0 || deg > 2 || d1 < -32768 || d1 > 32767;
    NodeLite& l = lite[v];
    l.head2 = r.head2;
    l.edge0 = r.edge0;
    l.meta = slow ? LITE_SLOW
                  : (r.len & 63u) | ((uint32_t)deg << 6) | ((uint32_t)((r.w0 >> 56) & 63u) << 8) |
                        ((uint32_t)(d1 & 0xFFFF) << 16);
  }
  });
  int st;
  if ((st = upload(ctx, ctx->lite, lite.data(), n, 1, "graph: 16-byte node records"))) return st;
  if ((st = upload(ctx, ctx->nodes, recs.data(), n, 1, "graph: 32-byte node records"))) return st;
  if ((st = upload(ctx, ctx->node_id, g->node_id, n, 1, "graph: node ids"))) return st;
  if ((st = upload(ctx, ctx->lab2, lab2.data(), lab2.size(), 0, "graph: 2-bit labels"))) return st;
  if ((st = upload(ctx, ctx->labn, labn.data(), labn.size(), 0, "graph: N mask"))) return st;
  if ((st = upload(ctx, ctx->edge_to, g->edge_to, n ? g->edge_off[n] : 0, 1, "graph: edge targets"))) return st;
  ctx->n_nodes = n;
  ctx->have_graph = true;
  ctx->id_base = n ? g->node_id[0] : 0;
  ctx->id_affine = true;
  for (uint64_t v = 0; v < n && ctx->id_affine; ++v) ctx->id_affine = g->node_id[v] == ctx->id_base + v;
  ctx->max_node_len = 0;
  ctx->node_len.resize(n);
  for (uint64_t v = 0; v < n; ++v) {
    ctx->node_len[v] = (uint32_t)(g->label_off[v + 1] - g->label_off[v]);
    ctx->max_node_len = std::max<uint64_t>(ctx->max_node_len, ctx->node_len[v]);
  }
  ctx->ids_sorted.release();
  if (!ctx->id_affine) {
    // the hit sorter orders by node id: keys hold a node's position among the sorted ids
    std::vector<uint64_t> ids(g->node_id, g->node_id + n);
    std::sort(ids.begin(), ids.end());
    if ((st = upload(ctx, ctx->ids_sorted, ids.data(), n, 1))) return st;
  }
  lkt_release(ctx);
  return loader_fence(ctx);
}

static int build_row_records(psigpu_ctx* ctx, uint32_t k);

// segment table of one part: (text start, node offset, external node id) per segment + a sentinel record at
// text_len, the directory, and the node rank of every segment
static int upload_segments(psigpu_ctx* ctx, const psigpu_index_view* x, const std::vector<uint64_t>& ids, DevBuf& seg,
                           DevBuf& seg_dir, DevBuf& seg_rank)
{
  std::vector<SegRec> segs;
  psigpu::resize_populated(segs, x->n_segs + 1);
  std::atomic<bool> foreign{ false };
  const uint64_t n_nodes = ctx->n_nodes;
  parallel_for(x->n_segs, 1u << 16, [&](uint64_t i0, uint64_t i1) {      // (300 M segments at whole-genome size)
    for (uint64_t i = i0; i < i1; ++i) {
      uint32_t v = x->seg_node[i];
      if (v != NO_NODE && v >= n_nodes) { foreign = true; v = NO_NODE; }
      segs[i] = SegRec{ x->seg_start[i], x->seg_noff[i], v == NO_NODE ? 0 : ids[v] };
    }
  });
  if (foreign) { ctx->err = "index does not belong to this graph"; return PSIGPU_ERR_ARG; }
  segs[x->n_segs] = SegRec{ x->seg_start[x->n_segs], 0, 0 };
  int st;
  if ((st = upload(ctx, seg, segs.data(), segs.size(), 1, "index: segment table"))) return st;
  if ((st = upload(ctx, seg_dir, x->seg_dir, x->n_dir, 1, "index: segment directory"))) return st;
  return upload(ctx, seg_rank, x->seg_node, x->n_segs, 1, "index: segment node ranks");
}

// one part's arrays -> the device; every array length follows from the part's text length
static int load_part(psigpu_ctx* ctx, const psigpu_index_view* m, uint32_t sa_rate, bool fm_ok, const std::vector<uint64_t>& ids,
                     psigpu_ctx::FmPart& fp)
{
  if (m->text_len == 0 || m->text_len >= 0xFFFFFFF0ull) { ctx->err = "text too long for the 32-bit index layout (or empty)"; return PSIGPU_ERR_ARG; }
  const bool has_fm = m->bwt_blocks != nullptr && m->n_blocks != 0;
  if (has_fm != fm_ok || (fm_ok && (m->n_blocks != m->text_len / BLOCK_SYMS + 1 || !m->exc_super || m->exc_shift > EXC_SUPER_SHIFT)) ||
      (!fm_ok && (m->n_exc || m->ftab_len)) || m->n_dir != (m->text_len >> DIR_SHIFT) + 1 ||
      m->n_samples != (m->text_len + sa_rate - 1) / sa_rate || (m->n_exc && (!m->exc_row || !m->exc_sa)) ||
      !m->sa_samples || !m->seg_start || !m->seg_dir || (!fm_ok && (sa_rate != 1 || !m->text4)) || m->ftab_len > 16) {
    ctx->err = "inconsistent index view";
    return PSIGPU_ERR_ARG;
  }
  {
    std::atomic<bool> bad_dir{ false };
    const uint64_t seg_lim = m->n_segs + (m->n_segs == 0);
    parallel_for(m->n_dir, 1u << 16, [&](uint64_t i0, uint64_t i1) {
      bool bad = false;
      for (uint64_t i = i0; i < i1; ++i) bad = bad || m->seg_dir[i] >= seg_lim;
      if (bad) bad_dir = true;
    });
    if (bad_dir) { ctx->err = "inconsistent index view"; return PSIGPU_ERR_ARG; }
  }
  int st;
  fp.release();
  if ((st = upload(ctx, fp.blocks, (const RankBlock*)m->bwt_blocks, fm_ok ? m->n_blocks : 0, 1, "index: rank blocks"))) return st;
  fp.exc_shift = fm_ok ? m->exc_shift : EXC_SUPER_SHIFT;
  if ((st = upload(ctx, fp.samples, m->sa_samples, m->n_samples, 1, "index: suffix array samples"))) return st;
  {
    // the exception rows and, behind them, the per-super-block counts (FMView::exc_super)
    const uint64_t n_super = fm_ok ? ((m->n_blocks - 1) >> m->exc_shift) + 1 : 0;
    HIPCHK(ctx, fp.exc_row.ensure((m->n_exc + n_super + 1) * 4 + 16));
    if (m->n_exc) HIPCHK(ctx, hipMemcpy(fp.exc_row.p, m->exc_row, m->n_exc * 4, hipMemcpyHostToDevice));
    if (n_super) HIPCHK(ctx, hipMemcpy(fp.exc_row.as<uint32_t>() + m->n_exc, m->exc_super, n_super * 4, hipMemcpyHostToDevice));
    if (!ab_no_pad_zero()) HIPCHK(ctx, hipMemsetAsync(fp.exc_row.as<uint32_t>() + m->n_exc + n_super, 0, 4, nullptr));
    if (!ab_load_hole()) HIPCHK(ctx, hipStreamSynchronize(nullptr));
    resident_forget(ctx, fp.exc_row);
    if (m->n_exc && !ab_load_hole()) { int rs = resident_note(ctx, "index: exception rows", fp.exc_row, fp.exc_row.p, m->exc_row, m->n_exc * 4); if (rs != PSIGPU_OK) return rs; }
  }
  if ((st = upload(ctx, fp.exc_sa, m->exc_sa, m->n_exc, 1, "index: exception positions"))) return st;
  if (m->text4) {
    if ((st = upload(ctx, fp.text4, m->text4, m->text_len / 16 + 2, 0, "index: 4-bit text"))) return st;
    fp.have_text4 = true;
  }
  if (m->ftab_len && m->ftab) {
    if ((st = upload(ctx, fp.ftab, m->ftab, 2ull << (2 * m->ftab_len), 0, "index: interval table"))) return st;
    fp.ftab_len = m->ftab_len;
  }
  if ((st = upload_segments(ctx, m, ids, fp.seg, fp.seg_dir, fp.seg_rank))) return st;
  fp.text_len = m->text_len; fp.n_exc = m->n_exc; fp.n_segs = m->n_segs;
  for (int i = 0; i < 4; ++i) fp.C[i] = m->C[i];
  return PSIGPU_OK;
}

int psigpu_load_index(psigpu_ctx* ctx, const psigpu_index_view* x)
{
  if (!ctx || !x) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (x->sa_rate == 0 || (x->sa_rate & (x->sa_rate - 1))) { ctx->err = "bad sa_rate"; return PSIGPU_ERR_ARG; }
  // nothing inconsistent reaches the kernels
  // (no rank blocks: a view that can only be tabulated -- k-mer table mode; this library's builder always makes them)
  const bool fm_ok = x->bwt_blocks != nullptr && x->n_blocks != 0;
  if (x->seed_len == 0 || x->seed_len > PSIGPU_MAX_SEED_LEN || (x->n_loci && (!x->loci_node || !x->loci_off))) {
    ctx->err = "inconsistent index view";
    return PSIGPU_ERR_ARG;
  }
  if (x->n_more_parts > PSIGPU_MAX_PARTS - 1 || (x->n_more_parts && !x->more_parts)) { ctx->err = "too many index parts"; return PSIGPU_ERR_ARG; }
  if (!ctx->have_graph) { ctx->err = "load the graph before the index"; return PSIGPU_ERR_STATE; }
  {
    std::atomic<bool> bad_locus{ false };
    const uint64_t n_nodes = ctx->n_nodes;
    const uint32_t* node_len = ctx->node_len.data();
    parallel_for(x->n_loci, 1u << 16, [&](uint64_t i0, uint64_t i1) {       // (1.3 G loci at whole-genome size)
      bool bad = false;
      for (uint64_t i = i0; i < i1; ++i) bad = bad || x->loci_node[i] >= n_nodes || x->loci_off[i] >= node_len[x->loci_node[i]];
      if (bad) bad_locus = true;
    });
    if (bad_locus) { ctx->err = "starting locus outside the graph: the index does not belong to this graph"; return PSIGPU_ERR_ARG; }
  }
  ctx->have_index = false;
  int st;
  std::vector<uint64_t> ids(ctx->n_nodes);
  if (ctx->n_nodes) HIPCHK(ctx, hipMemcpy(ids.data(), ctx->node_id.p, ctx->n_nodes * 8, hipMemcpyDeviceToHost));
  // (the checksums of the index arrays go with the arrays: every part is loaded again below)
  ctx->resident.erase(std::remove_if(ctx->resident.begin(), ctx->resident.end(),
                                     [](const psigpu_ctx::Resident& r) { return r.name.compare(0, 6, "index:") == 0; }), ctx->resident.end());
  while (ctx->parts.size() > 1) { ctx->parts.back()->release(); ctx->parts.pop_back(); }
  if ((st = load_part(ctx, x, x->sa_rate, fm_ok, ids, ctx->p0()))) return st;
  // further parts (an index whose text passes the 32-bit row limit): each a complete FM index of its own
  for (uint32_t pi = 0; pi < x->n_more_parts; ++pi) {
    const psigpu_index_view* m = &x->more_parts[pi];
    if (m->sa_rate != x->sa_rate) { ctx->err = "inconsistent index part"; return PSIGPU_ERR_ARG; }
    ctx->parts.emplace_back(new psigpu_ctx::FmPart);
    if ((st = load_part(ctx, m, x->sa_rate, fm_ok, ids, *ctx->parts.back()))) return st;
  }
  {
    // (pages first touched by the threads that fill them: 10 GB at whole-genome size)
    std::unique_ptr<uint2, void (*)(void*)> lc((uint2*)malloc((x->n_loci + 1) * sizeof(uint2)), free);
    if (!lc) { ctx->err = "out of host memory"; return PSIGPU_ERR_NOMEM; }
    uint2* lcp = lc.get();
    parallel_for(x->n_loci, 1u << 16, [&](uint64_t i0, uint64_t i1) {
      for (uint64_t i = i0; i < i1; ++i) lcp[i] = make_uint2(x->loci_node[i], x->loci_off[i]);
    });
    if ((st = upload(ctx, ctx->loci, lcp, x->n_loci, 1, "index: starting loci"))) return st;
  }
  // (the per-row records of the FM modes are made when an FM mode first answers a chunk: the default mode
  // never reads them, and at whole-genome size they are 70 GB)
  ctx->rows_tried = false;
  ctx->fm_ok = fm_ok;
  ctx->index_k = x->seed_len; ctx->sa_rate = x->sa_rate; ctx->context = x->context;
  ctx->n_paths = x->n_paths;
  ctx->n_loci = x->n_loci;
  ctx->have_index = true;
  lkt_release(ctx);
  pfx_release(ctx);
  ctx->auto_resolved = false;
  return loader_fence(ctx);
}

// Per-row records of the FM modes (SaRec for seed length k, and the located suffix array): derived from
// the suffix array, the text and the segment table that are already on the device.  Skipped (not an error)
// when they do not fit or do not apply.
static int build_row_records(psigpu_ctx* ctx, uint32_t k)
{
  static const bool env_no_sarec = getenv("PSIGPU_NO_SAREC") != nullptr;     // A/B (process-wide; per context: psigpu_set_tuning)
  const bool no_sarec = env_no_sarec || (ctx->tune & PSIGPU_TUNE_NO_ROWRECS);
  for (auto& pp : ctx->parts) {
    psigpu_ctx::FmPart& fp = *pp;
    fp.sarec_k = 0; fp.sarec.release();
    fp.ftabx_k = 0; fp.ftabx.release();
    fp.have_saloc = false; fp.saloc.release();
    const uint64_t n_rows = (fp.text_len + ctx->sa_rate - 1) / ctx->sa_rate;
    if (ctx->sa_rate == 1 && fp.have_text4 && fp.ftab_len && k >= fp.ftab_len && k - fp.ftab_len <= 29 && fp.n_segs && !no_sarec) {
      hipError_t e = fp.sarec.ensure(n_rows * sizeof(SaRec));
      if (e == hipSuccess) {
        k_build_sarec<<<(unsigned)((n_rows + 255) / 256), 256>>>(
            fp.samples.as<uint32_t>(), n_rows, k - fp.ftab_len, fp.seg.as<SegRec>(), fp.seg_rank.as<uint32_t>(),
            fp.seg_dir.as<uint32_t>(), fp.text4.as<uint64_t>(), fp.sarec.as<SaRec>());
        HIPCHK(ctx, hipDeviceSynchronize());
        fp.sarec_k = k;
        // the interval table with row l's record in its entries: 32 bytes per q-mer, when that is a small part of what is free
        static const bool env_no_ftabx = getenv("PSIGPU_NO_FTABX") != nullptr;      // A/B
        const uint64_t n_ent = 1ull << (2 * fp.ftab_len);
        size_t free_b = 0, total_b = 0;
        if (!env_no_ftabx && hipMemGetInfo(&free_b, &total_b) == hipSuccess && n_ent * sizeof(FtabX) * 4 < free_b &&
            fp.ftabx.ensure(n_ent * sizeof(FtabX)) == hipSuccess) {
          k_build_ftabx<<<(unsigned)((n_ent + 255) / 256), 256>>>(fp.ftab.as<uint2>(), n_ent, fp.sarec.as<SaRec>(), fp.ftabx.as<FtabX>());
          HIPCHK(ctx, hipDeviceSynchronize());
          fp.ftabx_k = k;
        } else {
          (void)hipGetLastError();
          fp.ftabx.release();
        }
      } else {
        (void)hipGetLastError();
        fp.sarec.release();
      }
    }
    if (ctx->sa_rate == 1 && fp.n_segs && !no_sarec && ctx->fm_ok) {
      // located suffix array (FM modes): only when it is a small part of what is free (the tables come later)
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (uint64_t)n_rows * 8 * 6 < free_b &&
          fp.saloc.ensure(n_rows * sizeof(uint2)) == hipSuccess) {
        k_build_saloc<<<(unsigned)((n_rows + 255) / 256), 256>>>(fp.samples.as<uint32_t>(), n_rows, fp.seg.as<SegRec>(),
                                                                 fp.seg_rank.as<uint32_t>(), fp.seg_dir.as<uint32_t>(),
                                                                 fp.saloc.as<uint2>());
        HIPCHK(ctx, hipDeviceSynchronize());
        fp.have_saloc = true;
      } else {
        (void)hipGetLastError();
        fp.saloc.release();
      }
    }
  }
  return PSIGPU_OK;
}

// the row records of every part, dropped (room for the k-mer table; psigpu_set_tuning)
static void drop_row_records(psigpu_ctx* ctx)
{
  for (auto& pp : ctx->parts) { pp->sarec.release(); pp->saloc.release(); pp->ftabx.release(); pp->sarec_k = pp->ftabx_k = 0; pp->have_saloc = false; }
  ctx->rows_tried = false;
}

static bool have_row_records(const psigpu_ctx* ctx)
{
  for (auto& pp : ctx->parts) if (pp->sarec_k != 0 || pp->have_saloc) return true;
  return false;
}

// The k-mer table built straight into its 16-byte slots (see k_pk_encode), over all parts of the index.
// `okeys` / `ovals`: the sorted (k-mer, locus) pairs.  Returns PSIGPU_ERR_NOMEM when it does not fit
// (kt_ready stays false).
static int build_kt_direct(psigpu_ctx* ctx, uint32_t k, const uint64_t* okeys, uint32_t* ovals, uint64_t n_off,
                           unsigned long long* d_cnt /* room for four counters */)
{
  const uint32_t n_parts = (uint32_t)ctx->parts.size();
  const bool dedup = ctx->gocc_thr == 0;
  ctx->kt_dedup = dedup;
  TmpBuf pk[PSIGPU_MAX_PARTS];
  PkParts parts{};
  parts.n_parts = n_parts;
  uint64_t rows_all = 0;
  for (uint32_t q = 0; q < n_parts; ++q) {
    PkPart& pt = parts.p[q];
    const psigpu_ctx::FmPart& fp = *ctx->parts[q];
    pt.n = fp.text_len; pt.sa = fp.samples.as<uint32_t>(); pt.seg = fp.seg.as<SegRec>();
    pt.seg_rank = fp.seg_rank.as<uint32_t>(); pt.seg_dir = fp.seg_dir.as<uint32_t>();
    const uint64_t* text4 = fp.text4.as<uint64_t>();
    rows_all += pt.n;
    if (pk[q].alloc((pt.n + 1) * 8) != hipSuccess) { (void)hipGetLastError(); return PSIGPU_ERR_NOMEM; }
    pt.pk = pk[q].as<uint64_t>();
    const unsigned grid = (unsigned)((pt.n + 255) / 256);
    k_pk_encode<<<grid, 256>>>(pt.sa, pt.n, k, text4, pk[q].as<uint64_t>());
    std::string serr;
    if (psigpu::gpu_running_max_u64(pk[q].as<uint64_t>(), pt.n, &serr) != PSIGPU_OK) { (void)hipGetLastError(); return PSIGPU_ERR_NOMEM; }
    k_pk_fix<<<grid, 256>>>(pt.sa, pt.n, k, text4, pk[q].as<uint64_t>());
  }
  const unsigned grid_off = (unsigned)((n_off + 255) / 256);
  // pass 1: how many k-mers need a 32-byte record, how many path k-mers there are, how many positions of
  // k-mers with several occurrences
  HIPCHK(ctx, hipMemset(d_cnt, 0, 32));
  for (uint32_t q = 0; q < n_parts; ++q)
    k_kt_direct_on<false><<<(unsigned)((parts.p[q].n + 255) / 256), 256>>>(parts, q, okeys, ovals, n_off, ctx->loci.as<uint2>(), nullptr, 0,
                                                                          nullptr, nullptr, d_cnt, dedup);
  if (n_off) k_kt_direct_off<false><<<grid_off, 256>>>(okeys, ovals, n_off, ctx->loci.as<uint2>(), parts, nullptr, 0, nullptr, d_cnt);
  unsigned long long h[3] = { 0, 0, 0 };
  HIPCHK(ctx, hipMemcpy(h, d_cnt, 24, hipMemcpyDeviceToHost));
  const uint64_t n_ext = h[0], n_on = h[1], n_pos = h[2];
  if (n_ext >= 0xFFFFFFF0ull || n_pos >= 0xFFFFFFF0ull) return PSIGPU_ERR_NOMEM;
  // slots: with looks that stay inside a sector (kt_next) the probe is as fast at load 1/3 as at 1/6 or 1/16
  // (chr22-like, one box, medians of three: 0.237 ms; 0.246 at load 0.5; 0.251 at 0.5 with plain linear
  // probing), so a table that stays small beside the free memory gets 3x the k-mers; otherwise load 0.5
  // when there is room, down to 0.85 when there is not
  uint64_t slots = 0;
  hipError_t e = hipErrorOutOfMemory;
  std::vector<uint64_t> pcts;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    if (const char* ev = getenv("PSIGPU_KT_PCT")) pcts.push_back(std::max<uint64_t>(101, strtoull(ev, nullptr, 10)));      // (experiments)
    if ((n_on + n_off) * 3 * sizeof(Slot16) <= free_b / 4) pcts.push_back(300);
    for (uint64_t pct : { 200ull, 160ull, 133ull, 118ull }) pcts.push_back(pct);
  }
  for (uint64_t pct : pcts) {
    slots = (std::max<uint64_t>(1024, (n_on + n_off) * pct / 100) + 3) & ~3ull;      // (an upper bound on the distinct k-mers; whole sectors)
    e = ctx->kt_ht.ensure(slots * sizeof(Slot16));
    if (e == hipSuccess) e = ctx->kt_ext.ensure((n_ext + 1) * sizeof(KmerSlot));
    if (e == hipSuccess) e = ctx->kt_onpos.ensure((n_pos + 1) * sizeof(uint2));
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    ctx->kt_ht.release(); ctx->kt_ext.release(); ctx->kt_onpos.release();
  }
  if (e != hipSuccess) return PSIGPU_ERR_NOMEM;
  HIPCHK(ctx, hipMemset(ctx->kt_ht.p, 0xFF, slots * sizeof(Slot16)));
  HIPCHK(ctx, hipMemset(d_cnt, 0, 32));
  for (uint32_t q = 0; q < n_parts; ++q)
    k_kt_direct_on<true><<<(unsigned)((parts.p[q].n + 255) / 256), 256>>>(parts, q, okeys, ovals, n_off, ctx->loci.as<uint2>(),
                                                                         ctx->kt_ht.as<Slot16>(), slots, ctx->kt_ext.as<KmerSlot>(),
                                                                         ctx->kt_onpos.as<uint2>(), d_cnt, dedup);
  if (n_off)
    k_kt_direct_off<true><<<grid_off, 256>>>(okeys, ovals, n_off, ctx->loci.as<uint2>(), parts, ctx->kt_ht.as<Slot16>(), slots,
                                             ctx->kt_ext.as<KmerSlot>(), d_cnt);
  HIPCHK(ctx, hipDeviceSynchronize());
  ctx->kt_ht_size = slots; ctx->kt_n_path_kmers = n_on; ctx->kt_n_ext = n_ext;
  ctx->kt_ready = true;
  (void)rows_all;
  return PSIGPU_OK;
}

// Builds the locus k-mer table for seed length k (see the comment above k_enum_compact).  Never
// fails the query: when the table cannot be built (memory, 32-bit entry index) the context
// stays with the query-time traverser and says why in lkt_note.
static int ensure_lkt(psigpu_ctx* ctx, uint32_t k, const GraphView& gv)
{
  if (ctx->lkt_ready && ctx->lkt_k == k) return PSIGPU_OK;
  if (ctx->lkt_failed && ctx->lkt_k == k) return PSIGPU_OK;
  lkt_release(ctx);
  ctx->lkt_k = k;
  auto give_up = [&](const std::string& why) {
    (void)hipGetLastError();
    lkt_release(ctx);
    ctx->lkt_failed = true;
    ctx->lkt_note = why;
    return PSIGPU_OK;
  };
#define LKT_TRY(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ == hipErrorOutOfMemory) return give_up("not enough device memory for the locus k-mer table"); \
    if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return PSIGPU_ERR_DEVICE; } \
  } while (0)
  const uint64_t n_loci = ctx->n_loci;
  const uint32_t walk_cap = ctx->walk_cap ? ctx->walk_cap : 256u;
  hipEvent_t e0, e1;
  HIPCHK(ctx, hipEventCreate(&e0)); HIPCHK(ctx, hipEventCreate(&e1));
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evg{ e0, e1 };
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipEventRecord(e0, nullptr));
  HIPCHK(ctx, ctx->w_ctr.ensure(sizeof(DevCounters)));
  DevCounters* ctr = ctx->w_ctr.as<DevCounters>();
  TmpBuf keys_a, vals_a, keys_b, spill_a, spill_b, total;
  uint64_t spill_cap = 1u << 22;                  // grows when a pass overflows it
  LKT_TRY(spill_a.alloc(spill_cap * sizeof(TravItem)));
  LKT_TRY(spill_b.alloc(spill_cap * sizeof(TravItem)));
  LKT_TRY(total.alloc(64));
  unsigned long long* d_dropped = (unsigned long long*)((char*)total.p + 8);
  TableView tb{};                                   // no seed table, no prefix filter: every walk
  DevCounters h{};
  // One enumeration pass: every k-walk of `n_roots` loci, up to `cap` walks per locus, into chunks
  // of (k-mer, root index) pairs.  status: 0 done, 1 gave up (why), 2 device error (ctx->err).
  struct Pass {
    TmpBuf walks, chunks, fill, chunk_off;
    uint64_t cap_chunks = 0, used_chunks = 0, n_pairs = 0, n_walks = 0;
    uint32_t cap = 0;
  };
  std::string why;
  if (ctx->query_mode == PSIGPU_MODE_TRAVERSE) {
    // Traverse mode: nothing about the LOCI is tabulated (graphs with too many k-walks per locus) -- the paths'
    // k-mers still are, one entry per path position whatever the graph looks like, so that the on-path phase
    // is one probe per seed as in the k-mer table mode and the traverser does the rest.  When the table does not
    // apply or fit, the FM index answers (PSIGPU_TUNE_NO_PATH_TABLE: always).
    spill_a.drop(); spill_b.drop();
    if (ctx->sa_rate == 1 && ctx->p0().have_text4 && ctx->n_paths && !(ctx->tune & PSIGPU_TUNE_NO_PATH_TABLE)) {
      int st = build_kt_direct(ctx, k, nullptr, nullptr, 0, d_dropped);
      if (st == PSIGPU_ERR_NOMEM) {
        (void)hipGetLastError();
        ctx->kt_ht.release(); ctx->kt_ext.release(); ctx->kt_onpos.release();
        ctx->lkt_note = "k-mer table of the paths does not fit the device: path k-mers stay with the FM index";
        st = PSIGPU_OK;
      }
      if (st != PSIGPU_OK) return st;
    }
    HIPCHK(ctx, hipEventRecord(e1, nullptr));
    HIPCHK(ctx, hipDeviceSynchronize());
    (void)hipEventElapsedTime(&ctx->lkt_build_ms, e0, e1);
    ctx->lkt_n_res = n_loci;
    ctx->lkt_ready = true;
    return PSIGPU_OK;
  }
  auto enumerate = [&](Pass& ps, const uint2* roots, uint64_t n_roots, uint32_t cap, uint64_t chunk_budget, bool retry) -> int {
#define PASS_TRY(call)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ == hipErrorOutOfMemory) { why = "not enough device memory for the locus k-mer table"; return 1; } \
    if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return 2; } \
  } while (0)
    ps.cap = cap;
    PASS_TRY(ps.walks.alloc(n_roots * 4 + 16));
    const uint32_t per_wave = (uint32_t)std::max<uint64_t>(256, (n_roots + 24575) / 24576);
    const uint64_t n_waves = (n_roots + per_wave - 1) / per_wave;
    ps.cap_chunks = chunk_budget ? chunk_budget : 2 * n_roots / CHUNK + n_waves + 4096;
    for (int attempt = 0, regrown = 0;; ++attempt) {
      if (ps.cap_chunks >= 0xFFFFFFF0ull) { why = "too many k-walks from the starting loci"; return 1; }
      PASS_TRY(ps.chunks.alloc(ps.cap_chunks * CHUNK * sizeof(ulonglong2)));
      PASS_TRY(ps.fill.alloc((ps.cap_chunks + 1) * 4));
      PASS_TRY(hipMemset(ps.fill.p, 0, (ps.cap_chunks + 1) * 4));
      PASS_TRY(hipMemset(ps.walks.p, 0, n_roots * 4 + 16));
      PASS_TRY(hipMemset(ctr, 0, sizeof(DevCounters)));
      EnumOut eo = { ps.chunks.as<ulonglong2>(), ps.fill.as<uint32_t>(), (uint32_t)ps.cap_chunks, ps.walks.as<uint32_t>(), cap };
      if (n_roots)
        k_traverse<true, uint64_t><<<(unsigned)n_waves, 64>>>(gv, tb, roots, n_roots, per_wave, nullptr, 0, spill_a.as<TravItem>(),
                                                    spill_cap, k, 0, nullptr, nullptr, 0, ctx->n_nodes, ctr, eo);
      PASS_TRY(hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
      TmpBuf* qin = &spill_a;
      TmpBuf* qout = &spill_b;
      bool spill_overflow = false;
      while (h.n_spill.v) {
        unsigned long long ns = h.n_spill.v;
        if (ns > spill_cap) {
          // the surplus was dropped: grow the queue and run the pass again
          if (spill_cap >= (1ull << 28) || ++regrown > 6) {
            why = "traverser spill queue overflow while enumerating the starting loci"; return 1;
          }
          spill_cap = std::min<uint64_t>(1ull << 28, std::max<uint64_t>(2 * spill_cap, ns + ns / 4));
          PASS_TRY(spill_a.alloc(spill_cap * sizeof(TravItem)));
          PASS_TRY(spill_b.alloc(spill_cap * sizeof(TravItem)));
          spill_overflow = true;
          break;
        }
        PASS_TRY(hipMemset(&ctr->n_spill.v, 0, 8));
        k_traverse<true, uint64_t><<<(unsigned)((ns + 63) / 64), 64>>>(gv, tb, roots, n_roots, 64, qin->as<TravItem>(), ns,
                                                            qout->as<TravItem>(), spill_cap, k, 0, nullptr, nullptr, 0,
                                                            ctx->n_nodes, ctr, eo);
        std::swap(qin, qout);
        PASS_TRY(hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
      }
      if (spill_overflow) { --attempt; continue; }
      if (h.n_chunks.v <= ps.cap_chunks) break;
      if (attempt || !retry) { why = "enumeration chunk overflow"; return 1; }
      ps.cap_chunks = h.n_chunks.v + 1024;
    }
    ps.used_chunks = h.n_chunks.v;
    ps.n_walks = h.n_kpaths.total();
    // chunk fills -> offsets, total number of pairs
    const uint64_t chunk_tiles = ps.cap_chunks / SCAN_TILE + 1;
    TmpBuf tiles;
    PASS_TRY(ps.chunk_off.alloc((ps.cap_chunks + 2) * 8));
    PASS_TRY(tiles.alloc(chunk_tiles * 8));
    k_scan_tiles<<<(unsigned)chunk_tiles, SCAN_THREADS>>>(ps.fill.as<uint32_t>(), ps.cap_chunks, tiles.as<uint64_t>());
    k_scan_sums<<<1, SCAN_THREADS>>>(tiles.as<uint64_t>(), chunk_tiles, total.as<uint64_t>());
    k_scan_final<<<(unsigned)chunk_tiles, SCAN_THREADS>>>(ps.fill.as<uint32_t>(), ps.cap_chunks, tiles.as<uint64_t>(),
                                                         ps.chunk_off.as<uint64_t>());
    PASS_TRY(hipMemcpy(&ps.n_pairs, total.p, 8, hipMemcpyDeviceToHost));
    return 0;
#undef PASS_TRY
  };
  auto count_over = [&](const Pass& ps, uint64_t n_roots, const uint2* roots, const uint32_t* ids_in, uint2* out,
                        uint32_t* ids_out, uint64_t* n) -> int {
    HIPCHK(ctx, hipMemset(d_dropped, 0, 8));
    if (n_roots)
      k_lkt_residual<<<(unsigned)((n_roots + 255) / 256), 256>>>(ps.walks.as<uint32_t>(), n_roots, ps.cap, roots, ids_in, out,
                                                                ids_out, d_dropped);
    HIPCHK(ctx, hipMemcpy(n, d_dropped, 8, hipMemcpyDeviceToHost));
    return PSIGPU_OK;
  };

  // pass 1: all starting loci, walk_cap walks each.  Loci over the cap: when they are few and the
  // cap is the default one, further passes give them far larger caps (a handful of dense sites
  // should not bring the per-chunk seed table and the traverser back for every chunk) under a
  // budget of pairs; what is still over after the last pass stays with the traverser.
  struct Tier { uint32_t cap; uint64_t max_loci, max_pairs; };
  const Tier tiers[] = { { walk_cap, ~0ull, 0 }, { 1u << 16, 65536, 64ull << 20 }, { 1u << 22, 1024, 256ull << 20 } };
  constexpr int N_TIERS = 3;
  Pass pass[N_TIERS];
  TmpBuf roots_buf[N_TIERS], ids_buf[N_TIERS];      // [t]: the loci of pass t (t >= 1) and their indices in ctx->loci
  int n_pass = 0;
  uint64_t n_walks_all = 0, n_res = 0, n_roots_cur = n_loci;
  const uint2* roots_cur = ctx->loci.as<uint2>();
  const uint32_t* ids_cur = nullptr;
  for (int t = 0; t < N_TIERS; ++t) {
    if (t > 0) {
      // loci of the previous pass that went over its cap
      const Pass& prev = pass[t - 1];
      if (prev.n_walks <= prev.n_pairs) { n_res = 0; break; }
      int st = count_over(prev, n_roots_cur, roots_cur, nullptr, nullptr, nullptr, &n_res);
      if (st != PSIGPU_OK) return st;
      if (n_res == 0 || n_res > tiers[t].max_loci || ctx->walk_cap != 0) break;
      LKT_TRY(roots_buf[t].alloc((n_res + 1) * sizeof(uint2)));
      LKT_TRY(ids_buf[t].alloc((n_res + 1) * 4));
      st = count_over(prev, n_roots_cur, roots_cur, ids_cur, roots_buf[t].as<uint2>(), ids_buf[t].as<uint32_t>(), &n_res);
      if (st != PSIGPU_OK) return st;
    }
    const uint2* r = t ? roots_buf[t].as<uint2>() : roots_cur;
    const uint64_t nr = t ? n_res : n_loci;
    int st = enumerate(pass[t], r, nr, tiers[t].cap, t ? tiers[t].max_pairs / CHUNK : 0, t == 0);
    if (st == 2) return PSIGPU_ERR_DEVICE;
    if (st == 1) {
      if (t == 0) return give_up(why);
      break;                                        // over the budget: the loci of this tier stay with the traverser
    }
    n_pass = t + 1;
    n_walks_all += pass[t].n_walks;
    if (t) { roots_cur = r; ids_cur = ids_buf[t].as<uint32_t>(); n_roots_cur = nr; }
  }
  uint64_t n_pairs = 0;
  for (int t = 0; t < n_pass; ++t) n_pairs += pass[t].n_pairs;
  if (n_pairs >= 0xFFFFFFF0ull) return give_up("more than 2^32 k-walks from the starting loci");
  LKT_TRY(keys_a.alloc((n_pairs + 1) * 8));
  LKT_TRY(vals_a.alloc((n_pairs + 1) * 4));
  HIPCHK(ctx, hipMemset(d_dropped, 0, 8));
  {
    uint64_t at = 0;
    for (int t = 0; t < n_pass; ++t) {
      Pass& ps = pass[t];
      if (ps.used_chunks)
        k_enum_compact<<<(unsigned)ps.used_chunks, 256>>>(ps.chunks.as<ulonglong2>(), ps.fill.as<uint32_t>(),
                                                         ps.chunk_off.as<uint64_t>(), (uint32_t)ps.cap_chunks,
                                                         ps.walks.as<uint32_t>(), ps.cap, k,
                                                         t ? ids_buf[t].as<uint32_t>() : nullptr,
                                                         keys_a.as<uint64_t>() + at, vals_a.as<uint32_t>() + at, d_dropped);
      at += ps.n_pairs;
    }
  }
  unsigned long long n_dropped = 0;
  HIPCHK(ctx, hipMemcpy(&n_dropped, d_dropped, 8, hipMemcpyDeviceToHost));
  // the loci that stay with the per-chunk traverser: over the cap of the last pass that ran
  {
    const Pass& last = pass[n_pass - 1];
    uint64_t n_left = 0;
    if (last.n_walks > last.n_pairs) {
      int st = count_over(last, n_roots_cur, roots_cur, nullptr, nullptr, nullptr, &n_left);
      if (st != PSIGPU_OK) return st;
      LKT_TRY(ctx->lkt_res.ensure((n_left + 1) * sizeof(uint2)));
      st = count_over(last, n_roots_cur, roots_cur, nullptr, ctx->lkt_res.as<uint2>(), nullptr, &n_left);
      if (st != PSIGPU_OK) return st;
    }
    n_res = n_left;
  }
  for (int t = 0; t < N_TIERS; ++t) {
    pass[t].chunks.drop(); pass[t].fill.drop(); pass[t].chunk_off.drop(); pass[t].walks.drop();
    roots_buf[t].drop(); ids_buf[t].drop();
  }
  spill_a.drop(); spill_b.drop();
  const uint64_t n_ent = n_pairs - n_dropped;
  const uint64_t* sorted_keys = keys_a.as<uint64_t>();
  uint32_t* sorted_vals = vals_a.as<uint32_t>();
  // the sorted loci stay: they are the locus runs the tables point into (LocusEnt)
  LKT_TRY(ctx->lkt_ent.ensure((n_pairs + 1) * sizeof(LocusEnt)));
  if (n_pairs) {
    LKT_TRY(keys_b.alloc((n_pairs + 1) * 8));
    std::string serr;
    int st = psigpu::gpu_sort_pairs_u64(keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(),
                                        ctx->lkt_ent.as<uint32_t>(), n_pairs, 2 * k + 1, &serr);
    if (st != PSIGPU_OK) return give_up("sorting the (k-mer, locus) pairs failed: " + serr);
    sorted_keys = keys_b.as<uint64_t>(); sorted_vals = ctx->lkt_ent.as<uint32_t>();
    keys_a.drop(); vals_a.drop();
  }
  // k-mer table mode: path k-mers and locus k-mers in one table of 16-byte slots (needs the whole suffix
  // array and the text on the device); when it does not fit, the 16-byte locus table below
  if (ctx->query_mode == PSIGPU_MODE_KMER_TABLE && ctx->sa_rate == 1 && ctx->p0().have_text4 && ctx->n_paths) {      // (load_part: a text in one part, a text in all)
    int st = build_kt_direct(ctx, k, sorted_keys, sorted_vals, n_ent, d_dropped);
    if (st == PSIGPU_ERR_NOMEM) {
      // not beside the per-row records of the FM modes (whole-genome indexes): give those up for the room;
      // they are made again if the table does not fit even then
      (void)hipGetLastError();
      ctx->kt_ht.release(); ctx->kt_ext.release();
      const bool had_rows = have_row_records(ctx);
      drop_row_records(ctx);
      st = had_rows ? build_kt_direct(ctx, k, sorted_keys, sorted_vals, n_ent, d_dropped) : PSIGPU_ERR_NOMEM;
      if (st == PSIGPU_ERR_NOMEM) {
        (void)hipGetLastError();
        ctx->kt_ht.release(); ctx->kt_ext.release();
        ctx->lkt_note = "k-mer table does not fit the device: path k-mers stay with the FM index";
        st = PSIGPU_OK;
      }
    }
    if (st != PSIGPU_OK) return st;
  }
  uint64_t ht_size = 1024;
  if (!ctx->kt_ready) {
    // load 0.5 when there is room, up to 0.8 when there is not (whole-genome graphs)
    hipError_t e = hipErrorOutOfMemory;
    for (uint64_t pct : { 200ull, 150ull, 125ull }) {
      ht_size = std::max<uint64_t>(1024, n_ent * pct / 100);
      e = ctx->lkt_ht.ensure(ht_size * sizeof(TableSlot));
      if (e != hipErrorOutOfMemory) break;
      (void)hipGetLastError();
    }
    LKT_TRY(e);
    HIPCHK(ctx, hipMemset(ctx->lkt_ht.p, 0xFF, ht_size * sizeof(TableSlot)));
    if (n_ent)
      k_lkt_insert<<<(unsigned)((n_ent + 255) / 256), 256>>>(sorted_keys, sorted_vals, ctx->loci.as<uint2>(), n_ent,
                                                             ctx->lkt_ht.as<TableSlot>(), ht_size);
  }
  HIPCHK(ctx, hipEventRecord(e1, nullptr));
  HIPCHK(ctx, hipDeviceSynchronize());
  (void)hipEventElapsedTime(&ctx->lkt_build_ms, e0, e1);
  ctx->lkt_ht_size = ht_size; ctx->lkt_n_ent = n_ent; ctx->lkt_n_res = n_res; ctx->lkt_n_walks = n_walks_all;
  ctx->lkt_ready = true;
#undef LKT_TRY
  return PSIGPU_OK;
}

static GraphView graph_view(const psigpu_ctx* ctx);

// The loci's PREFIX WALKS (k_traverse's pfx_roots): every walk of PFX_SHORT bases from every starting locus, with where it
// stands after its last base, in locus order.  A function of the graph and the loci alone, made once per index by the
// traverser itself in enumeration mode (seed length PFX_SHORT, no cap, prefix pairs), then ordered by locus (the node
// window a wave stages follows its first root).  When it cannot be made (memory, a flood of walks) the traverser starts
// from the loci as before.
static int ensure_pfx_roots(psigpu_ctx* ctx, const GraphView& gv)
{
  if (ctx->pfx_ready || ctx->pfx_failed) return PSIGPU_OK;
  const uint64_t n_loci = ctx->n_loci;
  auto give_up = [&](const char*) { (void)hipGetLastError(); pfx_release(ctx); ctx->pfx_failed = true; return PSIGPU_OK; };
#define PFX_TRY(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ == hipErrorOutOfMemory) return give_up("memory");                                   \
    if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return PSIGPU_ERR_DEVICE; } \
  } while (0)
  if (n_loci == 0 || n_loci >= 0xFFFFFFF0ull) return give_up("no loci");
  const auto t0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, ctx->w_ctr.ensure(sizeof(DevCounters)));
  DevCounters* ctr = ctx->w_ctr.as<DevCounters>();
  TmpBuf walks, chunks, fill, chunk_off, tiles, total, spill_a, spill_b;
  uint64_t spill_cap = 1u << 22;
  PFX_TRY(spill_a.alloc(spill_cap * sizeof(TravItem)));
  PFX_TRY(spill_b.alloc(spill_cap * sizeof(TravItem)));
  PFX_TRY(total.alloc(64));
  PFX_TRY(walks.alloc(n_loci * 4 + 16));
  const uint2* roots = ctx->loci.as<uint2>();
  const uint32_t per_wave = (uint32_t)std::max<uint64_t>(256, (n_loci + 24575) / 24576);
  const uint64_t n_waves = (n_loci + per_wave - 1) / per_wave;
  uint64_t cap_chunks = 2 * n_loci / CHUNK + n_waves + 4096;
  TableView tb{};
  DevCounters h{};
  for (int attempt = 0, regrown = 0;; ++attempt) {
    if (cap_chunks >= 0xFFFFFFF0ull || cap_chunks * CHUNK * sizeof(ulonglong2) > (64ull << 30)) return give_up("too many prefix walks");
    PFX_TRY(chunks.alloc(cap_chunks * CHUNK * sizeof(ulonglong2)));
    PFX_TRY(fill.alloc((cap_chunks + 1) * 4));
    PFX_TRY(hipMemset(fill.p, 0, (cap_chunks + 1) * 4));
    PFX_TRY(hipMemset(walks.p, 0, n_loci * 4 + 16));
    PFX_TRY(hipMemset(ctr, 0, sizeof(DevCounters)));
    EnumOut eo = { chunks.as<ulonglong2>(), fill.as<uint32_t>(), (uint32_t)cap_chunks, walks.as<uint32_t>(), 0xFFFFFFFFu, 1u };
    k_traverse<true, uint64_t><<<(unsigned)n_waves, 64>>>(gv, tb, roots, n_loci, per_wave, nullptr, 0, spill_a.as<TravItem>(), spill_cap,
                                                PFX_SHORT, 0, nullptr, nullptr, 0, ctx->n_nodes, ctr, eo);
    PFX_TRY(hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
    TmpBuf* qin = &spill_a;
    TmpBuf* qout = &spill_b;
    bool spill_overflow = false;
    while (h.n_spill.v) {
      const unsigned long long ns = h.n_spill.v;
      if (ns > spill_cap) {
        if (spill_cap >= (1ull << 28) || ++regrown > 6) return give_up("spill");
        spill_cap = std::min<uint64_t>(1ull << 28, std::max<uint64_t>(2 * spill_cap, ns + ns / 4));
        PFX_TRY(spill_a.alloc(spill_cap * sizeof(TravItem)));
        PFX_TRY(spill_b.alloc(spill_cap * sizeof(TravItem)));
        spill_overflow = true;
        break;
      }
      PFX_TRY(hipMemset(&ctr->n_spill.v, 0, 8));
      k_traverse<true, uint64_t><<<(unsigned)((ns + 63) / 64), 64>>>(gv, tb, roots, n_loci, 64, qin->as<TravItem>(), ns, qout->as<TravItem>(),
                                                          spill_cap, PFX_SHORT, 0, nullptr, nullptr, 0, ctx->n_nodes, ctr, eo);
      std::swap(qin, qout);
      PFX_TRY(hipMemcpy(&h, ctr, sizeof h, hipMemcpyDeviceToHost));
    }
    if (spill_overflow) { --attempt; continue; }
    if (h.n_chunks.v <= cap_chunks) break;
    if (attempt) return give_up("chunks");
    cap_chunks = h.n_chunks.v + 1024;
  }
  const uint64_t used_chunks = h.n_chunks.v;
  const uint64_t chunk_tiles = cap_chunks / SCAN_TILE + 1;
  PFX_TRY(chunk_off.alloc((cap_chunks + 2) * 8));
  PFX_TRY(tiles.alloc(chunk_tiles * 8));
  k_scan_tiles<<<(unsigned)chunk_tiles, SCAN_THREADS>>>(fill.as<uint32_t>(), cap_chunks, tiles.as<uint64_t>());
  k_scan_sums<<<1, SCAN_THREADS>>>(tiles.as<uint64_t>(), chunk_tiles, total.as<uint64_t>());
  k_scan_final<<<(unsigned)chunk_tiles, SCAN_THREADS>>>(fill.as<uint32_t>(), cap_chunks, tiles.as<uint64_t>(), chunk_off.as<uint64_t>());
  uint64_t n = 0;
  PFX_TRY(hipMemcpy(&n, total.p, 8, hipMemcpyDeviceToHost));
  spill_a.drop(); spill_b.drop(); walks.drop();
  if (n == 0 || n >= 0xFFFFFFF0ull) return give_up("no walks");
  TmpBuf raw, keys_a, keys_b, vals_a, vals_b;
  PFX_TRY(raw.alloc(n * 16));
  PFX_TRY(keys_a.alloc((n + 1) * 8)); PFX_TRY(keys_b.alloc((n + 1) * 8));
  PFX_TRY(vals_a.alloc((n + 1) * 4)); PFX_TRY(vals_b.alloc((n + 1) * 4));
  if (used_chunks)
    k_pfx_compact<<<(unsigned)used_chunks, 256>>>(chunks.as<ulonglong2>(), fill.as<uint32_t>(), chunk_off.as<uint64_t>(), (uint32_t)cap_chunks,
                                                 raw.as<uint4>(), keys_a.as<uint64_t>(), vals_a.as<uint32_t>());
  PFX_TRY(hipDeviceSynchronize());
  chunks.drop();
  {
    std::string err;
    int st = psigpu::gpu_sort_pairs_u64(keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(), vals_b.as<uint32_t>(), n, 32, &err);
    if (st == PSIGPU_ERR_NOMEM) return give_up("sort");
    if (st != PSIGPU_OK) { ctx->err = err; return st; }
  }
  PFX_TRY(ctx->pfx_roots.ensure((n + 64) * 16));
  k_pfx_gather<<<(unsigned)((n + 255) / 256), 256>>>(raw.as<uint4>(), vals_b.as<uint32_t>(), n, ctx->pfx_roots.as<uint4>());
  PFX_TRY(hipDeviceSynchronize());
  ctx->pfx_n = n;
  ctx->pfx_ready = true;
  ctx->pfx_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("PSIGPU_TRACE"))
    fprintf(stderr, "[psigpu] prefix walks of the starting loci: %llu walks of %u bases from %llu loci, %.1f ms\n", (unsigned long long)n, PFX_SHORT,
            (unsigned long long)n_loci, ctx->pfx_build_ms);
  return PSIGPU_OK;
#undef PFX_TRY
}

static GraphView graph_view(const psigpu_ctx* ctx)
{
  GraphView gv;
  gv.nodes = ctx->nodes.as<NodeRec>(); gv.lite = ctx->lite.as<NodeLite>(); gv.lab2 = ctx->lab2.as<uint64_t>(); gv.labn = ctx->labn.as<uint64_t>();
  gv.edge_to = ctx->edge_to.as<uint32_t>(); gv.node_id = ctx->node_id.as<uint64_t>();
  gv.id_base = ctx->id_base; gv.id_affine = ctx->id_affine;
  return gv;
}

// the look-back words of k_kmer_step for a call of at most `n_seeds` seeds: every word is either zero or carries the
// serial of the call that wrote it, so they are zeroed only when the allocation is new and when the 22-bit serial wraps
static int tilestate_ready(psigpu_ctx* ctx, uint64_t n_seeds, hipStream_t stream, bool may_grow)
{
  const size_t bytes = (n_seeds / KS_TILE + 66) * 8;
  if (!may_grow && ctx->w_tilestate.cap < bytes) return PSIGPU_ERR_NOMEM;
  HIPCHK(ctx, ctx->w_tilestate.ensure(bytes));
  if (ctx->tilestate_clean != ctx->w_tilestate.p || (ctx->serial & KS_SERIAL) == 0) {
    HIPCHK(ctx, hipMemsetAsync(ctx->w_tilestate.p, 0, ctx->w_tilestate.cap, stream));
    ctx->tilestate_clean = ctx->w_tilestate.p;
  }
  return PSIGPU_OK;
}
static bool fused_step_wanted(const psigpu_ctx* ctx)
{
  static const bool env_no_fused = getenv("PSIGPU_NO_FUSED") != nullptr;      // A/B
  return !env_no_fused && !ctx->opt_no_fused;
}

static int run_pipeline(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_read_off,
                        uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step, uint64_t rec_offset,
                        uint32_t flags, hipStream_t stream, uint64_t* n_hits_out, DevBuf* wire = nullptr,
                        const PackedIn* packed = nullptr, WireFmt wfmt = WireFmt{})
{
  // `packed`: d_bases is an array of 2-bit words (k_seed_pack<., true>), not ASCII
  // `wire`: the host entry's 16-byte records of the call's hits (k_hits_wire16), made behind the last kernel of the
  // call so that they are ready at its one host synchronisation
  // PSIGPU_SORT_UNIQUE here: the caller will sort -- when the hits come out seed by seed, order each seed's
  // hits in place before the counters go back, so that the answer to "was that enough?" comes with them
  const bool want_sorted = (flags & PSIGPU_SORT_UNIQUE) != 0;
  const uint32_t flags_in = flags;
  flags &= ~(PSIGPU_SORT_UNIQUE | PSIGPU_UNIFORM_READS);
  ctx->grouped_state = 0;
  if (step == 0) step = k;                       // src/psikt.cpp:469
  if (k == 0 || k > PSIGPU_MAX_SEED_LEN) { ctx->err = "seed length out of range (1..63)"; return PSIGPU_ERR_ARG; }
  if (!ctx->have_graph || !ctx->have_index) { ctx->err = "graph / index not loaded"; return PSIGPU_ERR_STATE; }
  resolve_auto_mode(ctx);
  if ((flags & PSIGPU_OFF_PATHS) && ctx->n_loci && ctx->index_k != k) {
    ctx->err = "starting loci were computed for a different seed length";
    return PSIGPU_ERR_ARG;
  }
  if (ctx->context != 0 && ctx->context < k) {
    ctx->err = "seed length should not be larger than context size";   // seed_finder.hpp:1434-1437
    return PSIGPU_ERR_CONTEXT;
  }
  if (n_reads >= 0xFFFFFFF0ull) { ctx->err = "too many reads in one chunk"; return PSIGPU_ERR_ARG; }
  // Per-kernel device times for psigpu_get_counters: HIP events on the streams the kernels run on.
  // Every record costs about 5 us of idle GPU, so a phase boundary that coincides with another one
  // is not recorded twice (the default mode records four per call).
#define EVREC(i, st_) HIPCHK(ctx, hipEventRecord(ctx->ev[i], st_))
  psigpu_counters& pc = ctx->last;
  memset(&pc, 0, sizeof pc);
  pc.n_reads = n_reads;
  pc.n_loci = ctx->n_loci;

  HIPCHK(ctx, ctx->w_ctr.ensure(sizeof(DevCounters)));
  HIPCHK(ctx, ctx->w_total.ensure(64));          // [0] seed count, [1] guess ratio
  DevCounters* ctr = ctx->w_ctr.as<DevCounters>();
  // the counters are zeroed by the first kernel of the call (k_seed_scan_tiles): a runtime memset is
  // a kernel launch of its own, three of them were 5 % of a step
  if (n_reads == 0) {
    HIPCHK(ctx, hipMemsetAsync(ctr, 0, sizeof(DevCounters), stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->w_total.p, 0, 16, stream));
  }
  EVREC(0, stream);

  // The seed table and the prefix bitmap are sized from an upper bound on the seed count
  // (every read of length L gives at most L / step + 1 seeds), so their reset can start now, on
  // the second stream, beside seeding -- it depends on nothing.
  // One stream: the traverser's kernels behind K1 / K2.  Running them beside each other on a second stream
  // (PSIGPU_OVERLAP=1, the arrangement of rounds 1-2) buys nothing on this part -- both sides wait for the same
  // random-access path: 1.91 ms against 1.93 per step in traverse mode -- and makes every event-bracketed kernel
  // time include the other stream's kernels.
  static const bool serial = getenv("PSIGPU_OVERLAP") == nullptr;
  // the second stream is made when a call first needs it (traverse mode): the runtime multiplexes
  // streams onto four hardware queues, and two busy streams on one queue serialise
  if (!ctx->stream2 && !serial && (flags & PSIGPU_OFF_PATHS) && ctx->n_loci)
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  const GraphView gv = graph_view(ctx);
  // Off-path hits: from the locus k-mer table (built on first use), the query-time traverser for
  // the loci the table leaves out -- or for all of them in PSIGPU_OFFPATH_TRAVERSE mode.
  const bool want_off = (flags & PSIGPU_OFF_PATHS) && ctx->n_loci && n_reads;
  // seeds of 32..63 bases are two words: answered by the FM index and the query-time traverser (the reference's
  // scheme); the tables of the other modes hold one-word k-mers
  const bool wide = k > PSIGPU_MAX_TABLE_SEED_LEN;
  const bool trav_mode = ctx->query_mode == PSIGPU_MODE_TRAVERSE;     // (its table holds the paths' k-mers only: ensure_lkt)
  const bool want_kt = ctx->index_k == k && n_reads && !wide &&
                       ((ctx->query_mode == PSIGPU_MODE_KMER_TABLE && (want_off || ((flags & PSIGPU_ON_PATHS) && ctx->n_paths))) ||
                        (trav_mode && (flags & PSIGPU_ON_PATHS) && ctx->n_paths && ctx->sa_rate == 1 &&
                         !(ctx->tune & PSIGPU_TUNE_NO_PATH_TABLE)));
  bool use_lkt = false, use_kt = false;
  if (((want_off && !trav_mode) || want_kt) && !wide) {
    const bool had = (ctx->lkt_ready || ctx->lkt_failed) && ctx->lkt_k == k;
    int st = ensure_lkt(ctx, k, gv);
    if (st != PSIGPU_OK) return st;
    use_lkt = ctx->lkt_ready && want_off && !trav_mode;
    use_kt = ctx->lkt_ready && ctx->kt_ready && want_kt;
    if (!had) EVREC(0, stream);      // a table build just ended: do not time it
  }
  // the per-row records of the FM search / locate kernels, the first time an FM mode answers on-path seeds
  if (!use_kt && n_reads && (flags & PSIGPU_ON_PATHS) && ctx->n_paths && ctx->fm_ok && !ctx->rows_tried) {
    int st = build_row_records(ctx, ctx->index_k);
    if (st != PSIGPU_OK) return st;
    ctx->rows_tried = true;
    EVREC(0, stream);
  }
  if (!ctx->fm_ok && n_reads && (flags & PSIGPU_ON_PATHS) && ctx->n_paths && !use_kt) {
    ctx->err = ctx->query_mode != PSIGPU_MODE_KMER_TABLE || ctx->index_k != k
                   ? "an index view without FM arrays is answered from the k-mer table only "
                     "(PSIGPU_MODE_KMER_TABLE, the index's seed length)"
                   : "the k-mer table of this index (no FM arrays) does not fit the device";
    return PSIGPU_ERR_STATE;
  }
  const uint2* trav_loci = use_lkt ? ctx->lkt_res.as<uint2>() : ctx->loci.as<uint2>();
  const uint64_t n_trav_loci = use_lkt ? ctx->lkt_n_res : ctx->n_loci;
  // traverse mode proper (every starting locus, every chunk), one-word seeds longer than the short prefix map: start from the
  // loci's prefix walks instead of the loci (PSIGPU_NO_PFX_ROOTS=1: from the loci, the A/B)
  static const bool env_no_pfx_roots = getenv("PSIGPU_NO_PFX_ROOTS") != nullptr;
  const uint4* pfx_roots = nullptr;
  uint64_t n_pfx_roots = 0;
  if (want_off && !use_lkt && !wide && k > PFX_SHORT && !env_no_pfx_roots && !ctx->opt_no_pfx_roots) {
    int st = ensure_pfx_roots(ctx, gv);
    if (st != PSIGPU_OK) return st;
    if (ctx->pfx_ready) { pfx_roots = ctx->pfx_roots.as<uint4>(); n_pfx_roots = ctx->pfx_n; EVREC(0, stream); }
  }
  pc.n_loci_traversed = (flags & PSIGPU_OFF_PATHS) ? n_trav_loci : 0;
  pc.n_locus_kmers = use_lkt ? ctx->lkt_n_ent : 0;
  pc.n_path_kmers = use_kt ? ctx->kt_n_path_kmers : 0;
  pc.ms_locus_table_build = (use_lkt || use_kt) ? ctx->lkt_build_ms : 0.f;
  const bool need_table = (flags & PSIGPU_OFF_PATHS) && n_trav_loci;
  const uint64_t seeds_ub = n_bases / step + n_reads;
  const uint32_t pfx_len = std::min<uint32_t>(k, PFX_LONG);
  const uint64_t pfx_words = ((1ull << (2 * pfx_len)) + 31) / 32;
  const bool use_pfx12 = need_table && k > PFX_SHORT;
  SeedBuckets sb;
  const char* sb_env = getenv("PSIGPU_SB_MAX");                         // (tests: 0 forces the one-region build on small chunks)
  const bool sb_parts = !wide && seeds_ub <= (sb_env ? strtoull(sb_env, nullptr, 10) : SB_MAX_SEEDS);     // partitioned build, or one region for the whole chunk
  sb.pb = sb_parts ? std::min<uint32_t>(SB_BASES, pfx_len) : 0u; sb.n_buckets = 1u << (2 * sb.pb);
  sb.n_wg = sb_parts ? (uint32_t)((seeds_ub + SB_TILE - 1) / SB_TILE) + 1 : 1u; sb.k = k;
  if (2 * seeds_ub >= 0xFFFFFFF0ull && need_table) { ctx->err = "too many seeds in one chunk for the traverser's seed table"; return PSIGPU_ERR_ARG; }
  const uint64_t sb_cnt = (uint64_t)sb.n_buckets * sb.n_wg;             // counters, bucket-major
  if (need_table && n_reads) {
    HIPCHK(ctx, ctx->w_ht.ensure((2 * seeds_ub + 16) * sizeof(TableSlot)));
    HIPCHK(ctx, ctx->w_pfx.ensure(pfx_words * 4 + 16));
    if (use_pfx12) HIPCHK(ctx, ctx->w_pfx12.ensure((1ull << (2 * PFX_SHORT)) / 8));
    HIPCHK(ctx, ctx->w_sb_cnt.ensure((sb_cnt + 1) * 4));
    HIPCHK(ctx, ctx->w_sb_off.ensure((sb_cnt + 2) * 8));
    HIPCHK(ctx, ctx->w_sb_tiles.ensure((sb_cnt / SCAN_TILE + 2) * 8));
    HIPCHK(ctx, ctx->w_sb_key.ensure((seeds_ub + 1) * 16));
  }

  // ---- K0: seeds ---------------------------------------------------------------------
  // PSIGPU_UNIFORM_READS: equal read lengths, claimed by the caller and checked by the packer
  UniformIn un{ 0, 0 };
  if ((flags_in & PSIGPU_UNIFORM_READS) && n_reads && !wide && n_bases % n_reads == 0 && n_bases / n_reads >= k &&
      n_bases / n_reads < (1ull << 31)) {
    un.len = (uint32_t)(n_bases / n_reads);
    un.spr = (un.len - k) / step + 1;
  }
  const bool uniform = un.spr != 0;
  if (uniform) {
    k_seed_init_uniform<<<1, 256, 0, stream>>>(ctr, ++ctx->serial, ctx->w_total.as<uint64_t>(), n_reads * (uint64_t)un.spr, un.len);
  } else
  if (n_reads) {
    uint64_t n_tiles = n_reads / SCAN_TILE + 1;     // covers index n_reads too
    HIPCHK(ctx, ctx->w_tiles.ensure(n_tiles * 8));
    HIPCHK(ctx, ctx->w_seed_off.ensure((n_reads + 1) * 8));
    k_seed_scan_tiles<<<(unsigned)n_tiles, SCAN_THREADS, 0, stream>>>(d_read_off, n_reads, k, step,
                                                                    ctx->w_tiles.as<uint64_t>(), ctr, ++ctx->serial);
    k_seed_scan_final<<<(unsigned)n_tiles, SCAN_THREADS, 0, stream>>>(d_read_off, n_reads, k, step,
                                                                    ctx->w_tiles.as<uint64_t>(),
                                                                    ctx->w_seed_off.as<uint64_t>(),
                                                                    ctx->w_total.as<uint64_t>(), ctr);
  }
  // No host round trip here: every buffer and grid below is sized by the upper bound, the kernels
  // read the true seed count from device memory (it comes back with the final counters).
  const uint64_t n_seeds = n_reads ? seeds_ub : 0;
  const uint64_t* d_params = ctx->w_total.as<uint64_t>();
  if (n_seeds >= 0xFFFFFFF0ull) { ctx->err = "too many seeds in one chunk"; return PSIGPU_ERR_ARG; }
  HIPCHK(ctx, ctx->w_seed_key.ensure((n_seeds + 1) * 8));
  HIPCHK(ctx, ctx->w_seed_info.ensure((n_seeds + 1) * 8));
  HIPCHK(ctx, ctx->w_seed_next.ensure((n_seeds + 1) * 4));
  if (wide) {
    HIPCHK(ctx, ctx->w_seed_wide.ensure((n_seeds + 1) * 16));
    HIPCHK(ctx, ctx->w_seed_pfx.ensure((n_seeds + 1) * 4));
  }
  // equal read lengths, every seed answered from the k-mer table: a seed's read and offset are its number divided -- the
  // (read, offset) array is neither written by the packer nor read by the emit kernel (16 of the step's ~100 bytes per seed)
  static const bool env_explicit_info = getenv("PSIGPU_EXPLICIT_INFO") != nullptr;      // A/B
  const bool implicit_info = uniform && use_kt && !need_table && !env_explicit_info;
  uint2* const d_seed_info = implicit_info ? nullptr : ctx->w_seed_info.as<uint2>();
  // the default step in one kernel (k_kmer_step): every seed answered by the k-mer table, no traverser behind it
  const bool fused = use_kt && n_seeds && !need_table && !wide && fused_step_wanted(ctx);
  if (fused) { int ts = tilestate_ready(ctx, n_seeds, stream, true); if (ts != PSIGPU_OK) return ts; }
  if (n_seeds && !fused) {
    const unsigned pgrid = (unsigned)std::min<uint64_t>((n_seeds + 256 * SP - 1) / (256 * SP), 256 * 32);
    if (uniform && packed)
      k_seed_pack<false, true, true><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, nullptr, n_reads, d_params, n_seeds, n_bases,
                                                                k, step, ctx->w_seed_key.as<uint64_t>(), d_seed_info, ctr,
                                                                nullptr, nullptr, 0, *packed, un);
    else if (uniform)
      k_seed_pack<false, false, true><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, nullptr, n_reads, d_params, n_seeds, n_bases,
                                                                 k, step, ctx->w_seed_key.as<uint64_t>(), d_seed_info, ctr,
                                                                 nullptr, nullptr, 0, PackedIn{ nullptr, 0, 0 }, un);
    else if (wide && packed)
      k_seed_pack<true, true><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, ctx->w_seed_off.as<uint64_t>(), n_reads, d_params, n_seeds, n_bases,
                                                         k, step, ctx->w_seed_key.as<uint64_t>(), ctx->w_seed_info.as<uint2>(), ctr,
                                                         ctx->w_seed_wide.as<u128>(), ctx->w_seed_pfx.as<uint32_t>(), pfx_len, *packed);
    else if (packed)
      k_seed_pack<false, true><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, ctx->w_seed_off.as<uint64_t>(), n_reads, d_params, n_seeds, n_bases,
                                                          k, step, ctx->w_seed_key.as<uint64_t>(), ctx->w_seed_info.as<uint2>(), ctr,
                                                          nullptr, nullptr, 0, *packed);
    else if (wide)
      k_seed_pack<true><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, ctx->w_seed_off.as<uint64_t>(), n_reads, d_params, n_seeds, n_bases,
                                                   k, step, ctx->w_seed_key.as<uint64_t>(), ctx->w_seed_info.as<uint2>(), ctr,
                                                   ctx->w_seed_wide.as<u128>(), ctx->w_seed_pfx.as<uint32_t>(), pfx_len);
    else
      k_seed_pack<false><<<pgrid, 256, 0, stream>>>(d_bases, d_read_off, ctx->w_seed_off.as<uint64_t>(), n_reads, d_params, n_seeds, n_bases,
                                                    k, step, ctx->w_seed_key.as<uint64_t>(), ctx->w_seed_info.as<uint2>(), ctr,
                                                    nullptr, nullptr, 0);
  }
  // the host entry answers a chunk in ~10 sub-batches whose kernels take 0.1 ms each: two of the five event records of a
  // default-mode call (each ~5 us of idle GPU) are left out there -- the per-kernel times of such a call are then 0, its
  // total and its sort time stay (PSIGPU_TRACE keeps all)
  const bool lean_ev = wire != nullptr && !ctx->trace_call;      // (the host entry reads PSIGPU_TRACE once per call)
  if (!lean_ev) EVREC(1, stream);
  static const bool env_no_verify = getenv("PSIGPU_NO_VERIFY") != nullptr;   // A/B: LF steps only
  const bool no_verify = env_no_verify || (ctx->tune & PSIGPU_TUNE_NO_VERIFY);
  auto fm_view = [&](const psigpu_ctx::FmPart& fp) {
    FMView fm;
    fm.blocks = fp.blocks.as<uint4>();
    fm.exc_row = fp.exc_row.as<uint32_t>();
    fm.exc_shift = (uint16_t)fp.exc_shift;
    fm.n_exc = (uint32_t)fp.n_exc;
    fm.n = (uint32_t)fp.text_len;
    for (int i = 0; i < 4; ++i) fm.C[i] = (uint32_t)fp.C[i];
    fm.ftab = fp.ftab_len ? fp.ftab.as<uint2>() : nullptr;
    fm.ftab_len = (uint16_t)fp.ftab_len;
    fm.text4 = (fp.have_text4 && !no_verify) ? fp.text4.as<uint64_t>() : nullptr;
    fm.sa = ctx->sa_rate == 1 ? fp.samples.as<uint32_t>() : nullptr;
    fm.sarec = (fp.sarec_k == k && !no_verify) ? fp.sarec.as<SaRec>() : nullptr;
    return fm;
  };
  auto map_view = [&](const psigpu_ctx::FmPart& fp, const FMView& fm) {
    MapView mv;
    mv.samples = fp.samples.as<uint32_t>(); mv.sa_rate = ctx->sa_rate;
    mv.exc_sa = fp.exc_sa.as<uint32_t>();
    mv.seg = fp.seg.as<SegRec>(); mv.seg_dir = fp.seg_dir.as<uint32_t>();
    mv.sarec = fm.sarec; mv.sarec_rem = k - fp.ftab_len;
    mv.node_id = ctx->node_id.as<uint64_t>(); mv.id_base = ctx->id_base; mv.id_affine = ctx->id_affine;
    mv.loci = ctx->loci.as<uint2>();
    mv.on_pos = ctx->kt_onpos.as<uint2>();
    mv.saloc = fp.have_saloc ? fp.saloc.as<uint2>() : nullptr;
    return mv;
  };
  TableView tb;
  tb.ht = ctx->w_ht.as<TableSlot>();
  tb.boff = ctx->w_sb_off.as<uint64_t>(); tb.n_wg = sb.n_wg; tb.pb = sb.pb;
  tb.seed_next = ctx->w_seed_next.as<uint32_t>();
  tb.seed_info = ctx->w_seed_info.as<uint2>();
  tb.seed_wide = wide ? ctx->w_seed_wide.p : nullptr;
  // PSIGPU_NO_PFX (diagnostic): no pruning, so n_kpaths counts every k-walk from the starting loci
  const bool no_pfx = getenv("PSIGPU_NO_PFX") != nullptr;
  tb.pfx12 = (use_pfx12 && !no_pfx) ? ctx->w_pfx12.as<uint32_t>() : nullptr;
  tb.pfx_bits = (need_table && !no_pfx) ? ctx->w_pfx.as<uint32_t>() : nullptr; tb.pfx_len = pfx_len;
  const bool kprobe = use_kt && n_seeds;               // k-mer table: the seed's interval and its loci in one probe
  static const bool env_res16 = getenv("PSIGPU_RES16") != nullptr;
  const bool res8 = kprobe && !env_res16 && !ctx->opt_res16 && ctx->max_node_len < (1ull << R8_NOFF_BITS);      // 8 bytes of probe results per seed
  const bool on_paths = (flags & PSIGPU_ON_PATHS) && ctx->n_paths && n_seeds && !kprobe;   // FM index (K1)
  const bool off_paths = need_table && n_seeds;        // query-time traverser
  const bool probe = use_lkt && n_seeds && !kprobe;    // locus k-mer table (16-byte slots) beside the FM index
  uint64_t spill_cap = ctx->spill_cap;

  // On-path work (K1 -> scan -> K2) runs on the caller's stream, the traverser (K4) beside it
  // on the context's second stream (both are latency-bound).  On-path hits land at scan-given
  // offsets; the traverser's chunks are packed behind them by k_chunk_compact.  Buffers are
  // sized from the previous call / a guess and the pass is retried once on overflow.
  hipStream_t s2 = ctx->stream2;
  uint64_t cap = std::max<uint64_t>(ctx->hits_cap_hint, 4 * n_seeds + (1u << 16));
  uint64_t cap_chunks = std::max<uint64_t>(ctx->chunks_cap_hint, n_seeds / CHUNK * 2 + 32768 + 1024);
  // the FM modes search every part of the index: per-seed K1 -> K2 arrays and per-wave totals per part
  const uint32_t n_fm = on_paths ? (uint32_t)ctx->parts.size() : 1u;
  const uint64_t seed_stride = n_seeds + 16;
  HIPCHK(ctx, ctx->w_iv_lo.ensure(n_fm * seed_stride * 4));
  HIPCHK(ctx, ctx->w_iv_cnt.ensure(n_fm * seed_stride * 4));
  HIPCHK(ctx, ctx->w_iv_aux.ensure(n_fm * seed_stride * 4));
  DevCounters& h = *reinterpret_cast<DevCounters*>(ctx->h_pinned);
  uint64_t& true_seeds = *reinterpret_cast<uint64_t*>((char*)ctx->h_pinned + sizeof(DevCounters));
  memset(&h, 0, sizeof h);
  true_seeds = 0;
  uint64_t total_hits = 0;
  constexpr int MAX_ATTEMPTS = 6;
  for (int attempt = 0; attempt < MAX_ATTEMPTS; ++attempt) {
    if (off_paths) {
      HIPCHK(ctx, ctx->w_spill_a.ensure(spill_cap * sizeof(TravItemT<u128>)));       // (room for either item type)
      HIPCHK(ctx, ctx->w_spill_b.ensure(spill_cap * sizeof(TravItemT<u128>)));
    }
    HIPCHK(ctx, ctx->w_hits.ensure((cap + 1) * sizeof(psigpu_hit)));
    psigpu_hit* d_hits = ctx->w_hits.as<psigpu_hit>();
    const uint64_t chunk_tiles = cap_chunks / SCAN_TILE + 1;
    if (off_paths) {
      HIPCHK(ctx, ctx->w_chunks.ensure(cap_chunks * CHUNK * sizeof(psigpu_hit)));
      HIPCHK(ctx, ctx->w_chunk_fill.ensure((cap_chunks + 1) * 4));
      HIPCHK(ctx, ctx->w_chunk_off.ensure((cap_chunks + 2) * 8));
      HIPCHK(ctx, ctx->w_chunk_tiles.ensure(chunk_tiles * 8));
      HIPCHK(ctx, hipMemsetAsync(ctx->w_chunk_fill.p, 0, (cap_chunks + 1) * 4, stream));
    }
    if (off_paths || on_paths) EVREC(3, stream);          // fork point of the second stream / start of K1
    pc.traverse_launches = 0;
    pc.n_spilled = 0;
    // the seeds "index" (table + prefix bitmaps) is only needed by the traverser
    auto launch_table = [&](hipStream_t ts) -> int {
      EVREC(2, ts);
      if (!sb_parts) {
        const uint32_t m = (uint32_t)(2 * seeds_ub);
        FillJob fa = { ctx->w_ht.as<uint4>(), m, 0xFFFFFFFFu };       // key = invalid, val, dup = NIL
        FillJob fb = { nullptr, 0, 0u };
        FillJob fc = { ctx->w_pfx.as<uint4>(), (pfx_words * 4 + 15) / 16, 0u };
        k_fill3<<<2048, 256, 0, ts>>>(fa, fb, fc);
        k_table_insert<<<(unsigned)((n_seeds + 255) / 256), 256, 0, ts>>>(
            ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, ctx->w_ht.as<TableSlot>(), m, ctx->w_seed_next.as<uint32_t>(), k,
            ctx->w_pfx.as<uint32_t>(), pfx_len, ctx->w_sb_off.as<uint64_t>(), wide ? ctx->w_seed_pfx.as<uint32_t>() : nullptr);
        if (use_pfx12)
          k_pfx_derive<<<(1u << (2 * PFX_SHORT)) / 32 / 256, 256, 0, ts>>>(ctx->w_pfx.as<uint32_t>(), pfx_len, ctx->w_pfx12.as<uint32_t>());
        EVREC(6, ts);
        return PSIGPU_OK;
      }
      // partition the seeds by their leading bases (count, scan, scatter), then one workgroup per bucket
      const size_t lds = sb.n_buckets * 4;
      k_sb_count<<<sb.n_wg, 256, lds, ts>>>(ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, sb, ctx->w_sb_cnt.as<uint32_t>());
      const uint64_t tiles = sb_cnt / SCAN_TILE + 1;
      k_scan_tiles<<<(unsigned)tiles, SCAN_THREADS, 0, ts>>>(ctx->w_sb_cnt.as<uint32_t>(), sb_cnt, ctx->w_sb_tiles.as<uint64_t>());
      k_scan_sums<<<1, SCAN_THREADS, 0, ts>>>(ctx->w_sb_tiles.as<uint64_t>(), tiles, ctx->w_sb_tiles.as<uint64_t>() + tiles);
      k_scan_final<<<(unsigned)tiles, SCAN_THREADS, 0, ts>>>(ctx->w_sb_cnt.as<uint32_t>(), sb_cnt, ctx->w_sb_tiles.as<uint64_t>(),
                                                           ctx->w_sb_off.as<uint64_t>());
      k_sb_scatter<<<sb.n_wg, 256, lds, ts>>>(ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, sb, ctx->w_sb_off.as<uint64_t>(),
                                             ctx->w_sb_key.as<ulonglong2>(), ctx->w_seed_next.as<uint32_t>());
      if ((1u << (2 * (pfx_len - sb.pb))) < 32)              // (prefix maps of short seeds: buckets share words)
        HIPCHK(ctx, hipMemsetAsync(ctx->w_pfx.p, 0, pfx_words * 4, ts));
      k_sb_build<<<sb.n_buckets, 256, 0, ts>>>(ctx->w_sb_key.as<ulonglong2>(), ctx->w_sb_off.as<uint64_t>(), sb,
                                              ctx->w_ht.as<TableSlot>(), ctx->w_seed_next.as<uint32_t>(), ctx->w_pfx.as<uint32_t>(), pfx_len,
                                              use_pfx12 ? ctx->w_pfx12.as<uint32_t>() : nullptr);
      EVREC(6, ts);
      return PSIGPU_OK;
    };
    auto launch_traverse = [&](hipStream_t ts) -> int {
      // ~96 waves per CU over the launch keeps the tail short and the atomics few
      const uint64_t n_roots_l = pfx_roots ? n_pfx_roots : n_trav_loci;
      const uint32_t per_wave = (uint32_t)std::max<uint64_t>(256, (n_roots_l + 24575) / 24576);
      uint64_t n_waves = (n_roots_l + per_wave - 1) / per_wave;
      if (wide)
        k_traverse<false, u128><<<(unsigned)n_waves, 64, 0, ts>>>(
            gv, tb, trav_loci, n_trav_loci, per_wave,
            nullptr, 0, ctx->w_spill_a.as<TravItemT<u128>>(), spill_cap, k, rec_offset,
            ctx->w_chunks.as<psigpu_hit>(), ctx->w_chunk_fill.as<uint32_t>(), (uint32_t)cap_chunks, ctx->n_nodes, ctr,
            EnumOut{});
      else
      k_traverse<false, uint64_t><<<(unsigned)n_waves, 64, 0, ts>>>(
          gv, tb, trav_loci, n_roots_l, per_wave,
          nullptr, 0, ctx->w_spill_a.as<TravItem>(), spill_cap, k, rec_offset,
          ctx->w_chunks.as<psigpu_hit>(), ctx->w_chunk_fill.as<uint32_t>(), (uint32_t)cap_chunks, ctx->n_nodes, ctr,
          EnumOut{}, pfx_roots);
      ++pc.traverse_launches;
      EVREC(7, ts);
      return PSIGPU_OK;
    };
    if (off_paths && !serial) {
      int st;
      HIPCHK(ctx, hipStreamWaitEvent(s2, ctx->ev[3], 0));
      if ((st = launch_table(s2)) != PSIGPU_OK) return st;
      if ((st = launch_traverse(s2)) != PSIGPU_OK) return st;
    }
    if (on_paths || probe || kprobe) {
      uint32_t thr = ctx->gocc_thr ? ctx->gocc_thr : 0xFFFFFFFFu;
      // one contiguous seed range per wave, 16 seeds per round; the same split in K1, the table probe and K2
      uint64_t n_waves = std::min<uint64_t>(WAVES_MAX, (n_seeds + 15) / 16);
      n_waves = (n_waves + 3) / 4 * 4;
      uint32_t per_wave = (uint32_t)(((n_seeds + n_waves - 1) / n_waves + 63) / 64 * 64);
      unsigned grid = (unsigned)(n_waves / 4);
      const uint64_t tiles_stride = n_waves + 2;             // + the part's range of output slots
      HIPCHK(ctx, ctx->w_iv_tiles.ensure(n_fm * tiles_stride * 8));
      HIPCHK(ctx, ctx->w_iv_tiles_off.ensure((n_waves + 1) * 8));
      // K1 -> K2 per-seed arrays: five more beside iv_lo / iv_cnt / iv_aux (on_node, on_noff per part; the
      // locus-table probe's three once)
      HIPCHK(ctx, ctx->w_seedout.ensure((2 * n_fm + 3) * seed_stride * 4));
      if (kprobe) HIPCHK(ctx, ctx->w_seedres.ensure((n_seeds + 16) * 16));
      auto seed_out = [&](uint32_t p) {
        SeedOut so;
        so.iv_lo = ctx->w_iv_lo.as<uint32_t>() + p * seed_stride; so.iv_cnt = ctx->w_iv_cnt.as<uint32_t>() + p * seed_stride;
        so.iv_aux = ctx->w_iv_aux.as<uint32_t>() + p * seed_stride;
        so.on_node = ctx->w_seedout.as<uint32_t>() + 2 * p * seed_stride; so.on_noff = so.on_node + seed_stride;
        so.off_first = ctx->w_seedout.as<uint32_t>() + 2 * n_fm * seed_stride; so.off_cnt = so.off_first + seed_stride;
        so.off_noff = so.off_cnt + seed_stride;
        return so;
      };
      auto tiles_of = [&](uint32_t p) { return ctx->w_iv_tiles.as<uint64_t>() + p * tiles_stride; };
      if (attempt) EVREC(10, stream);
      if (attempt == 0) {
        // K1: lane per seed when the interval table + text verification can finish a seed without
        // LF steps (the locus-table probe rides along); quad per seed otherwise, and for the seeds
        // the direct kernel defers
        static const bool env_no_direct = getenv("PSIGPU_NO_DIRECT") != nullptr;   // A/B: quad kernel only
        const bool no_direct = env_no_direct || (ctx->tune & PSIGPU_TUNE_NO_DIRECT);
        LktView lk = { nullptr, 0, nullptr };
        if (probe) lk = LktView{ ctx->lkt_ht.as<TableSlot>(), ctx->lkt_ht_size, ctx->lkt_ent.as<LocusEnt>() };
        bool probed = false;
        // a gocc threshold counts a k-mer's occurrences in ALL parts: with several, K1 reports every part's count
        // and k_parts_combine applies the threshold to the sum
        const bool combine = on_paths && n_fm > 1;
        const uint32_t thr_k1 = combine ? 0xFFFFFFFFu : thr;
        if (kprobe && fused) {
          const KmerTableView kt = { ctx->kt_ht.as<Slot16>(), ctx->kt_ht_size, ctx->kt_ext.as<KmerSlot>() };
          const MapView mv = map_view(ctx->p0(), fm_view(ctx->p0()));
          const unsigned sgrid = (unsigned)((n_seeds + KS_TILE - 1) / KS_TILE);
          const PackedIn pk0 = packed ? *packed : PackedIn{ nullptr, 0, 0 };
          const uint32_t serial22 = (uint32_t)(ctx->serial & KS_SERIAL);
#define STEP_ARGS d_bases, d_read_off, ctx->w_seed_off.as<uint64_t>(), n_reads, d_params, n_seeds, n_bases, k, step, pk0, un, kt, mv, \
                  ctx->lkt_ent.as<LocusEnt>(), (flags & PSIGPU_ON_PATHS) != 0, want_off && use_lkt, thr, rec_offset, d_hits, cap, \
                  ctx->w_tilestate.as<uint64_t>(), serial22, ctr, env_plain_stores
          if (uniform && packed) k_kmer_step<true, true><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
          else if (uniform) k_kmer_step<false, true><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
          else if (packed) k_kmer_step<true, false><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
          else k_kmer_step<false, false><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
#undef STEP_ARGS
          probed = true;
        } else if (kprobe) {
          KmerTableView kt = { ctx->kt_ht.as<Slot16>(), ctx->kt_ht_size, ctx->kt_ext.as<KmerSlot>() };
          if (res8)
            k_kmer_probe<true><<<grid, 256, 0, stream>>>(kt, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave,
                                                         (flags & PSIGPU_ON_PATHS) != 0, want_off && use_lkt, thr, ctx->w_seedres.as<uint4>(),
                                                         ctx->w_iv_tiles.as<uint64_t>(), ctx->w_iv_tiles_off.as<uint64_t>(), ctr);
          else
          k_kmer_probe<false><<<grid, 256, 0, stream>>>(kt, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave,
                                                 (flags & PSIGPU_ON_PATHS) != 0, want_off && use_lkt, thr, ctx->w_seedres.as<uint4>(),
                                                 ctx->w_iv_tiles.as<uint64_t>(), ctx->w_iv_tiles_off.as<uint64_t>(), ctr);
          probed = true;
        } else if (on_paths) {
          pc.search_launches = 0;
          for (uint32_t p = 0; p < n_fm; ++p) {
            const FMView fm = fm_view(*ctx->parts[p]);
            const SeedOut so = seed_out(p);
            const bool direct = fm.ftab != nullptr && fm.sarec != nullptr && !no_direct && !wide;      // (row records verify <= 29 bases)
            if (direct) {
              HIPCHK(ctx, ctx->w_defer.ensure((n_seeds + 1) * 4));
              if (p) HIPCHK(ctx, hipMemsetAsync(&ctr->n_defer.v, 0, 8, stream));
              const bool ride = probe && p == 0;            // the locus-table probe rides in the first part's kernel
              k_fm_search_direct<<<grid, 256, 0, stream>>>(
                  fm, ctx->parts[p]->ftabx_k == k ? ctx->parts[p]->ftabx.as<FtabX>() : nullptr,
                  ride ? lk : LktView{ nullptr, 0, nullptr }, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave, k, thr_k1, so,
                  tiles_of(p), ride ? ctx->w_iv_tiles_off.as<uint64_t>() : nullptr, ctx->w_defer.as<uint32_t>(), ctr);
              k_fm_search<true, uint64_t><<<256, 256, 0, stream>>>(fm, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave, k, thr_k1,
                                                   so.iv_lo, so.iv_cnt, so.iv_aux, tiles_of(p), ctr,
                                                   ctx->w_defer.as<uint32_t>(), &ctr->n_defer.v);
              pc.search_launches += 2;
              probed = probed || ride;
            } else {
              if (wide)
                k_fm_search<false, u128><<<grid, 256, 0, stream>>>(fm, ctx->w_seed_wide.as<u128>(), d_params, n_seeds, per_wave, k, thr_k1,
                                                                  so.iv_lo, so.iv_cnt, so.iv_aux, tiles_of(p), ctr, nullptr, nullptr);
              else
              k_fm_search<false, uint64_t><<<grid, 256, 0, stream>>>(fm, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave, k, thr_k1,
                                                    so.iv_lo, so.iv_cnt, so.iv_aux, tiles_of(p), ctr, nullptr, nullptr);
              pc.search_launches += 1;
            }
          }
          if (combine) {
            HIPCHK(ctx, hipMemsetAsync(&ctr->n_live, 0, sizeof(StripedCounter), stream));      // (K1 counted per part)
            k_parts_combine<<<grid, 256, 0, stream>>>(ctx->w_iv_cnt.as<uint32_t>(), seed_stride, n_fm, thr, d_params, n_seeds, per_wave,
                                                      ctx->w_iv_tiles.as<uint64_t>(), tiles_stride, ctr);
          }
        } else {
          HIPCHK(ctx, hipMemsetAsync(ctx->w_iv_cnt.p, 0, (n_seeds + 1) * 4, stream));
          HIPCHK(ctx, hipMemsetAsync(ctx->w_iv_aux.p, 0, (n_seeds + 1) * 4, stream));
          HIPCHK(ctx, hipMemsetAsync(ctx->w_iv_tiles.p, 0, (n_waves + 1) * 8, stream));
        }
        if (!kprobe) EVREC(10, stream);
        if (probe && !probed) {
          k_lkt_probe<<<grid, 256, 0, stream>>>(lk, ctx->w_seed_key.as<uint64_t>(), d_params, n_seeds, per_wave, seed_out(0),
                                                ctx->w_iv_tiles_off.as<uint64_t>());
        }
        // per-wave totals -> first output slot of every wave; total on-path hits, total K2 output: per part, each
        // part's hits behind the parts before it (the k-mer table mode's emit kernel does this itself, from the raw totals)
        if (!kprobe)
          for (uint32_t p = 0; p < n_fm; ++p)
            k_wave_offsets<<<1, 1024, 0, stream>>>(tiles_of(p), (probe && p == 0) ? ctx->w_iv_tiles_off.as<uint64_t>() : nullptr, n_waves,
                                                   (uint64_t*)&ctr->n_hits_on.v, (uint64_t*)&ctr->n_hits_tab.v, p != 0);
      }
      if (!(lean_ev && kprobe && !off_paths) && !fused) EVREC(4, stream);
      const LocusEnt* oe = (probe || kprobe) ? ctx->lkt_ent.as<LocusEnt>() : nullptr;
      if (fused) {
        // (the records are out already)
      } else if (kprobe) {
        const FMView fm0 = fm_view(ctx->p0());
        const uint32_t thr_e = ctx->gocc_thr ? ctx->gocc_thr : 0xFFFFFFFFu;
        if (res8)
          k_kmer_emit<true><<<grid, 256, 0, stream>>>(map_view(ctx->p0(), fm0), ctx->w_seedres.as<uint4>(), ctx->kt_ext.as<KmerSlot>(), oe,
                                                      ctx->w_iv_tiles.as<uint64_t>(), ctx->w_iv_tiles_off.as<uint64_t>(), d_params,
                                                      n_seeds, per_wave, d_seed_info, rec_offset, d_hits, cap, ctr,
                                                      (flags & PSIGPU_ON_PATHS) != 0, want_off && use_lkt, thr_e, un.spr, step, env_plain_stores);
        else
        k_kmer_emit<false><<<grid, 256, 0, stream>>>(map_view(ctx->p0(), fm0), ctx->w_seedres.as<uint4>(), ctx->kt_ext.as<KmerSlot>(), oe,
                                              ctx->w_iv_tiles.as<uint64_t>(), ctx->w_iv_tiles_off.as<uint64_t>(), d_params,
                                              n_seeds, per_wave, d_seed_info, rec_offset, d_hits, cap, ctr,
                                              (flags & PSIGPU_ON_PATHS) != 0, want_off && use_lkt, thr_e, un.spr, step, env_plain_stores);
      } else {
        if (ctx->sa_rate != 1) {
          HIPCHK(ctx, ctx->w_hit_a.ensure((cap + 1) * 8));
          HIPCHK(ctx, ctx->w_hit_seed.ensure((cap + 1) * 4));
        }
        for (uint32_t p = 0; p < n_fm; ++p) {
          const FMView fm = fm_view(*ctx->parts[p]);
          const MapView mv = map_view(*ctx->parts[p], fm);
          const SeedOut so = seed_out(p);
          const bool with_table = probe && p == 0;             // the locus table's hits follow the first part's
          if (ctx->sa_rate == 1)
            k_fm_locate_direct<<<grid, 256, 0, stream>>>(mv, so, with_table, oe, tiles_of(p), d_params, n_seeds,
                                                         per_wave, ctx->w_seed_info.as<uint2>(), rec_offset, d_hits, cap);
          else {
            // walks (decoupled quads) -> 12 bytes per hit -> records
            k_fm_walk<<<grid, 256, 0, stream>>>(fm, mv.samples, mv.sa_rate, mv.exc_sa, so.iv_lo, so.iv_cnt, with_table ? so.off_cnt : nullptr,
                                                tiles_of(p), d_params, n_seeds, per_wave,
                                                ctx->w_hit_a.as<uint64_t>(), ctx->w_hit_seed.as<uint32_t>(), cap, ctr);
            k_hits_resolve<<<2048, 256, 0, stream>>>(mv, ctx->w_hit_a.as<uint64_t>(), ctx->w_hit_seed.as<uint32_t>(), so.iv_cnt,
                                                     so.off_first, so.off_cnt, so.off_noff, oe, tiles_of(p) + n_waves,
                                                     ctx->w_seed_info.as<uint2>(), rec_offset, d_hits, cap);
          }
        }
      }
    } else {
      EVREC(10, stream);
      EVREC(4, stream);
    }
    const bool k2_ends_call = kprobe && !off_paths;      // nothing between K2 and the end of the call
    if (!k2_ends_call) EVREC(5, stream);
    if (off_paths && serial) {
      int st;
      if ((st = launch_table(stream)) != PSIGPU_OK) return st;
      if ((st = launch_traverse(stream)) != PSIGPU_OK) return st;
    }
    if (off_paths && !serial) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev[7], 0));   // join
    if (off_paths) {      // the traverser's spill count and chunk count are needed on the host
      k_publish<<<1, 256, 0, stream>>>(reinterpret_cast<const uint4*>(ctr), reinterpret_cast<uint4*>(ctx->h_pinned_dev), (uint32_t)(sizeof(DevCounters) / 16));
      HIPCHK(ctx, hipStreamSynchronize(stream));
    }
    bool spill_overflow = false;
    if (off_paths && h.n_spill.v) {
      // drain the traverser's spill queue (only dense / high-degree regions ever spill)
      DevBuf* qin = &ctx->w_spill_a;
      DevBuf* qout = &ctx->w_spill_b;
      unsigned long long ns = h.n_spill.v;
      while (ns) {
        if (ns > spill_cap) {
          // more partial walks than the queue holds: the surplus was dropped, so this attempt is void;
          // grow the queue (for this context, from now on) and run the traverser phase again
          if (spill_cap >= (1ull << 30)) { ctx->err = "traverser spill queue overflow"; return PSIGPU_ERR_NOMEM; }
          spill_cap = std::min<uint64_t>(1ull << 30, std::max<uint64_t>(2 * spill_cap, ns + ns / 4));
          ctx->spill_cap = spill_cap;
          spill_overflow = true;
          break;
        }
        pc.n_spilled += ns;
        HIPCHK(ctx, hipMemsetAsync(&ctr->n_spill.v, 0, 8, stream));
        const uint32_t pw = 64;
        if (wide)
          k_traverse<false, u128><<<(unsigned)((ns + pw - 1) / pw), 64, 0, stream>>>(
              gv, tb, trav_loci, n_trav_loci, pw,
              qin->as<TravItemT<u128>>(), ns, qout->as<TravItemT<u128>>(), spill_cap, k, rec_offset,
              ctx->w_chunks.as<psigpu_hit>(), ctx->w_chunk_fill.as<uint32_t>(), (uint32_t)cap_chunks, ctx->n_nodes, ctr,
              EnumOut{});
        else
        k_traverse<false, uint64_t><<<(unsigned)((ns + pw - 1) / pw), 64, 0, stream>>>(
            gv, tb, trav_loci, n_trav_loci, pw,
            qin->as<TravItem>(), ns, qout->as<TravItem>(), spill_cap, k, rec_offset,
            ctx->w_chunks.as<psigpu_hit>(), ctx->w_chunk_fill.as<uint32_t>(), (uint32_t)cap_chunks, ctx->n_nodes, ctr,
            EnumOut{});
        ++pc.traverse_launches;
        std::swap(qin, qout);
        k_publish<<<1, 256, 0, stream>>>(reinterpret_cast<const uint4*>(ctr), reinterpret_cast<uint4*>(ctx->h_pinned_dev), (uint32_t)(sizeof(DevCounters) / 16));
        HIPCHK(ctx, hipStreamSynchronize(stream));
        ns = h.n_spill.v;
      }
    }
    bool overflow = spill_overflow;
    if (off_paths) {
      if (h.n_chunks.v > cap_chunks) { overflow = true; cap_chunks = h.n_chunks.v + h.n_chunks.v / 8 + 1024; }
      else if (!spill_overflow) {
        // pack the chunks behind the on-path hits
        k_scan_tiles<<<(unsigned)chunk_tiles, SCAN_THREADS, 0, stream>>>(ctx->w_chunk_fill.as<uint32_t>(), cap_chunks,
                                                                       ctx->w_chunk_tiles.as<uint64_t>());
        k_scan_sums<<<1, SCAN_THREADS, 0, stream>>>(ctx->w_chunk_tiles.as<uint64_t>(), chunk_tiles,
                                                    (uint64_t*)&ctr->n_hits_off.v);
        k_scan_final<<<(unsigned)chunk_tiles, SCAN_THREADS, 0, stream>>>(ctx->w_chunk_fill.as<uint32_t>(), cap_chunks,
                                                                       ctx->w_chunk_tiles.as<uint64_t>(),
                                                                       ctx->w_chunk_off.as<uint64_t>());
        unsigned nblk = (unsigned)std::max<uint64_t>(1, h.n_chunks.v);
        k_chunk_compact<<<nblk, 256, 0, stream>>>(ctx->w_chunks.as<psigpu_hit>(), ctx->w_chunk_fill.as<uint32_t>(),
                                                ctx->w_chunk_off.as<uint64_t>(), (uint32_t)cap_chunks,
                                                &ctr->n_hits_tab.v, d_hits, cap);
      }
    }
    const bool fix_groups = want_sorted && !off_paths && cap != 0 && n_fm == 1;      // (several parts: hits come out part by part)
    if (fix_groups) {
      EVREC(11, stream);
      int fs = HitSorter::fix_grouped(d_hits, cap, &ctr->n_hits_tab.v, (uint64_t*)&ctr->not_grouped.v, stream, &ctx->err);
      if (fs != PSIGPU_OK) return fs;
    }
    EVREC(8, stream);
    unsigned long long* h_wflag = reinterpret_cast<unsigned long long*>((char*)ctx->h_pinned + sizeof(DevCounters) + 32);
    ctx->wire_used = 0;
    if (wire && cap && wfmt.bytes) {
      HIPCHK(ctx, wire->ensure((cap + 1) * 16));
      if (wfmt.bytes == 8) {
        *h_wflag = 0;
        k_hits_wire8<<<2048, 256, 0, stream>>>(d_hits, &ctr->n_hits_tab.v, &ctr->n_hits_off.v, 0, cap, ctx->id_base, rec_offset, wfmt,
                                               wire->as<uint64_t>(),
                                               reinterpret_cast<unsigned long long*>((char*)ctx->h_pinned_dev + sizeof(DevCounters) + 32));
      } else
        k_hits_wire16<<<2048, 256, 0, stream>>>(d_hits, &ctr->n_hits_tab.v, &ctr->n_hits_off.v, 0, cap, ctx->id_base, rec_offset,
                                                wire->as<uint4>());
      ctx->wire_used = wfmt.bytes;
    }
    k_publish<<<1, 256, 0, stream>>>(reinterpret_cast<const uint4*>(ctr), reinterpret_cast<uint4*>(ctx->h_pinned_dev), (uint32_t)(sizeof(DevCounters) / 16));
    HIPCHK(ctx, hipStreamSynchronize(stream));
    if (ctx->wire_used == 8 && *h_wflag) {
      // a field did not fit (a read longer than the bits the sub-batch left it): 16-byte records of this call's hits
      // instead -- one more kernel and synchronisation, once; the context stays with 16 bytes
      ctx->wire8_overflowed = true;
      k_hits_wire16<<<2048, 256, 0, stream>>>(d_hits, &ctr->n_hits_tab.v, &ctr->n_hits_off.v, 0, cap, ctx->id_base, rec_offset,
                                              wire->as<uint4>());
      HIPCHK(ctx, hipStreamSynchronize(stream));
      ctx->wire_used = 16;
    }
    if (n_reads && h.serial.v != ctx->serial) {
      // what came back is not this call's counter block (never seen alone; the check is there because a stale count is
      // exactly what one record missing / two extra under GPU sharing would look like): counted, reported, and the block
      // is fetched again with a runtime copy
      ++ctx->stale_handbacks;
      if (getenv("PSIGPU_TRACE")) fprintf(stderr, "[psigpu] stale counter hand-back: serial %llu, expected %llu\n", h.serial.v, ctx->serial);
      HIPCHK(ctx, hipMemcpy(&h, ctr, sizeof(DevCounters), hipMemcpyDeviceToHost));
      if (h.serial.v != ctx->serial) { ctx->err = "the counters of the call did not come back from the device"; return PSIGPU_ERR_DEVICE; }
    }
    if (uniform && h.not_uniform.v) {
      // the reads are not all of one length after all: the whole call again, the general way
      ++ctx->uniform_refuted;
      return run_pipeline(ctx, d_bases, d_read_off, n_reads, n_bases, k, step, rec_offset, flags_in & ~PSIGPU_UNIFORM_READS, stream,
                          n_hits_out, wire, packed, wfmt);
    }
    ctx->grouped_state = fix_groups ? (h.not_grouped.v ? 2 : 1) : 0;
    true_seeds = h.n_seeds_true.v;
    ctx->last_max_read_len = 0;
    for (int i = 0; i < STRIPES; ++i) ctx->last_max_read_len = std::max<uint64_t>(ctx->last_max_read_len, h.max_read_len.s[i].v);
    if (true_seeds > n_seeds) { ctx->err = "n_bases does not cover the reads"; return PSIGPU_ERR_ARG; }
    total_hits = h.n_hits_tab.v + h.n_hits_off.v;
    static const bool env_debug = getenv("PSIGPU_DEBUG") != nullptr;
    if (env_debug)
      fprintf(stderr, "[psigpu] attempt %d: seeds %llu of %llu, hits on %llu tab %llu off %llu, chunks %llu of %llu, spill %llu, cap %llu (kprobe %d on %d off %d probe %d)\n",
              attempt, (unsigned long long)true_seeds, (unsigned long long)n_seeds, h.hits_on(), h.n_hits_tab.v, h.n_hits_off.v, h.n_chunks.v,
              (unsigned long long)cap_chunks, h.n_spill.v, (unsigned long long)cap, (int)kprobe, (int)on_paths, (int)off_paths, (int)probe);
    if (total_hits > cap) { overflow = true; cap = total_hits + total_hits / 16 + 1024; }
    if (!overflow) break;
    if (fused) {
      // the one kernel of the step wrote what fitted and counted the rest: the call again with room for all of it
      ctx->hits_cap_hint = std::max<uint64_t>(ctx->hits_cap_hint, cap);
      return run_pipeline(ctx, d_bases, d_read_off, n_reads, n_bases, k, step, rec_offset, flags_in, stream, n_hits_out, wire, packed, wfmt);
    }
    if (attempt == MAX_ATTEMPTS - 1) { ctx->err = "hit buffer / spill queue overflow"; return PSIGPU_ERR_NOMEM; }
    HIPCHK(ctx, hipMemsetAsync(&ctr->n_kpaths, 0, sizeof(StripedCounter), stream));
    HIPCHK(ctx, hipMemsetAsync(&ctr->n_spill.v, 0, 8, stream));
    HIPCHK(ctx, hipMemsetAsync(&ctr->n_chunks.v, 0, 8, stream));
    HIPCHK(ctx, hipMemsetAsync(&ctr->n_hits_off.v, 0, 8, stream));
  }
  ctx->hits_cap_hint = std::max<uint64_t>(ctx->hits_cap_hint, total_hits + total_hits / 8);
  ctx->chunks_cap_hint = std::max<uint64_t>(ctx->chunks_cap_hint, h.n_chunks.v + h.n_chunks.v / 8 + 1024);
  pc.n_seeds = true_seeds;
  pc.n_seeds_valid = h.n_seeds_valid.total();
  pc.n_seeds_on_path = h.n_live.total();
  pc.n_hits_on_path = h.hits_on();
  pc.n_hits_off_path = (h.n_hits_tab.v - h.hits_on()) + h.n_hits_off.v;
  pc.n_hits = total_hits;
  pc.n_kpaths = h.n_kpaths.total();
  pc.n_lf_steps = h.n_lf_steps.total();
  pc.n_rows_verified = h.n_rows_verified.total();
  pc.n_locate_steps = h.n_locate_steps.total();
  static const bool env_debug2 = getenv("PSIGPU_DEBUG") != nullptr;
  if (env_debug2) fprintf(stderr, "[psigpu] dbg0 %llu dbg1 %llu chunks %llu spilled %llu\n", h.dbg0.v, h.dbg1.v, h.n_chunks.v, (unsigned long long)pc.n_spilled);
  auto ms = [&](int a, int b) { float t = 0; (void)hipEventElapsedTime(&t, ctx->ev[a], ctx->ev[b]); return t; };
  const bool lean_k = lean_ev && kprobe && !off_paths;          // (events 1 and 4 were not recorded)
  pc.ms_pack = lean_ev ? 0.f : ms(0, 1); pc.ms_table = off_paths ? ms(2, 6) : 0.f;
  pc.ms_search = on_paths ? ms(3, 10) : 0.f;          // K1
  // table probe + the scan of the per-wave totals
  pc.ms_probe = lean_k ? 0.f : kprobe ? (lean_ev ? 0.f : fused ? ms(1, ctx->grouped_state ? 11 : 8) : ms(1, 4)) : (probe ? ms(10, 4) : 0.f);
  pc.ms_locate = (lean_k || fused) ? 0.f : (on_paths || probe || kprobe) ? ((kprobe && !off_paths) ? ms(4, ctx->grouped_state ? 11 : 8) : ms(4, 5)) : 0.f;
  pc.fused_step = fused ? 1u : 0u;
  pc.ms_traverse = off_paths ? ms(6, 7) : 0.f;        // runs beside K1/K2 on the second stream
  pc.ms_total = ms(0, 8);
  if (ctx->grouped_state) { pc.ms_sort = ms(11, 8); pc.ms_total -= pc.ms_sort; }     // (added back by the caller with the sort's time)
  *n_hits_out = total_hits;
  // (the host entry may keep two sub-batches of such calls in flight: enqueue_default)
  if (kprobe && !off_paths && !on_paths && !probe && n_fm == 1) {
    ctx->fast_k = k; ctx->fast_flags = flags; ctx->fast_on = (flags & PSIGPU_ON_PATHS) != 0; ctx->fast_off = want_off && use_lkt;
  } else ctx->fast_k = 0;

  return PSIGPU_OK;
}

// ------------------------------------------------------------------------------------
// The default mode's kernels of ONE sub-batch of the host entry, queued WITHOUT a host synchronisation (round 4).  The host
// entry answers a chunk in ~7 sub-batches of ~0.1 ms of kernels each; run_pipeline ends every one with a synchronisation,
// and between the end of a sub-batch's last kernel and the start of the next one's first the device idled for what the
// host needs to wake up, read the counters, queue the transfer out and launch again (~75 us per sub-batch, a third of a
// 2-ms call).  Here the kernels of sub-batch i + 1 are in the queue before the host waits for sub-batch i: the per-call
// workspace is shared (the stream orders the kernels), what the HOST reads afterwards is per sub-batch in flight -- the
// hit buffer, the wire buffer, a mapped block for the counters, an event.  Only for calls that a run_pipeline call before
// them has shown to need nothing but the default mode's kernels (ctx->fast_k); anything unusual found when a sub-batch is
// finished -- more hits than its buffer holds, a seed with more hits than the in-place ordering takes, a wire field too
// narrow, reads that are not of one length after all -- hands the rest of the chunk to the synchronous loop.
// ------------------------------------------------------------------------------------
struct FastArgs {
  const char* d_in; const PackedIn* pk; const uint64_t* d_off;
  uint64_t nr, nb; uint32_t k, step; uint64_t rec_base;
  bool want_sort, claim_uniform;
  WireFmt wf; DevBuf* wire; uint64_t cap; int slot;
};

static int enqueue_default(psigpu_ctx* ctx, const FastArgs& a, hipStream_t stream, unsigned long long* serial_out, bool* uniform_out)
{
  psigpu_ctx::FastSlot& fs = ctx->fast[a.slot];
  DevCounters* ctr = ctx->w_ctr.as<DevCounters>();
  const uint32_t k = a.k, step = a.step;
  const uint64_t n_seeds = a.nb / step + a.nr;                     // upper bound: grids and buffers (sized by the caller)
  const uint64_t* d_params = ctx->w_total.as<uint64_t>();
  HIPCHK(ctx, hipEventRecord(fs.begin, stream));
  UniformIn un{ 0, 0 };
  if (a.claim_uniform && a.nr && a.nb % a.nr == 0 && a.nb / a.nr >= k && a.nb / a.nr < (1ull << 31)) {
    un.len = (uint32_t)(a.nb / a.nr);
    un.spr = (un.len - k) / step + 1;
  }
  const bool uniform = un.spr != 0;
  *uniform_out = uniform;
  const unsigned long long serial = ++ctx->serial;
  *serial_out = serial;
  if (uniform)
    k_seed_init_uniform<<<1, 256, 0, stream>>>(ctr, serial, ctx->w_total.as<uint64_t>(), a.nr * (uint64_t)un.spr, un.len);
  else {
    const uint64_t n_tiles = a.nr / SCAN_TILE + 1;
    k_seed_scan_tiles<<<(unsigned)n_tiles, SCAN_THREADS, 0, stream>>>(a.d_off, a.nr, k, step, ctx->w_tiles.as<uint64_t>(), ctr, serial);
    k_seed_scan_final<<<(unsigned)n_tiles, SCAN_THREADS, 0, stream>>>(a.d_off, a.nr, k, step, ctx->w_tiles.as<uint64_t>(),
                                                                    ctx->w_seed_off.as<uint64_t>(), ctx->w_total.as<uint64_t>(), ctr);
  }
  const uint32_t thr = ctx->gocc_thr ? ctx->gocc_thr : 0xFFFFFFFFu;
  psigpu_hit* d_hits = fs.hits.as<psigpu_hit>();
  MapView mv;
  {
    const psigpu_ctx::FmPart& fp = ctx->p0();
    mv.samples = fp.samples.as<uint32_t>(); mv.sa_rate = ctx->sa_rate;
    mv.exc_sa = fp.exc_sa.as<uint32_t>();
    mv.seg = fp.seg.as<SegRec>(); mv.seg_dir = fp.seg_dir.as<uint32_t>();
    mv.sarec = nullptr; mv.sarec_rem = k - fp.ftab_len;
    mv.node_id = ctx->node_id.as<uint64_t>(); mv.id_base = ctx->id_base; mv.id_affine = ctx->id_affine;
    mv.loci = ctx->loci.as<uint2>();
    mv.on_pos = ctx->kt_onpos.as<uint2>();
    mv.saloc = fp.have_saloc ? fp.saloc.as<uint2>() : nullptr;
  }
  const KmerTableView kt = { ctx->kt_ht.as<Slot16>(), ctx->kt_ht_size, ctx->kt_ext.as<KmerSlot>() };
  const LocusEnt* oe = ctx->lkt_ent.as<LocusEnt>();
  // (the look-back words must be there already: nothing grows under a chunk in flight -- the caller checked)
  const bool fused = fused_step_wanted(ctx);
  if (fused) { int ts = tilestate_ready(ctx, n_seeds, stream, false); if (ts != PSIGPU_OK) { ctx->err = "look-back words not sized for this sub-batch"; return ts; } }
  if (fused) {
    const unsigned sgrid = (unsigned)((n_seeds + KS_TILE - 1) / KS_TILE);
    const PackedIn pkf = a.pk ? *a.pk : PackedIn{ nullptr, 0, 0 };
    const uint32_t serial22 = (uint32_t)(serial & KS_SERIAL);
#define STEP_ARGS a.d_in, a.d_off, ctx->w_seed_off.as<uint64_t>(), a.nr, d_params, n_seeds, a.nb, k, step, pkf, un, kt, mv, oe, ctx->fast_on, \
                  ctx->fast_off, thr, a.rec_base, d_hits, a.cap, ctx->w_tilestate.as<uint64_t>(), serial22, ctr, env_plain_stores
    if (uniform && a.pk) k_kmer_step<true, true><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
    else if (uniform) k_kmer_step<false, true><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
    else if (a.pk) k_kmer_step<true, false><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
    else k_kmer_step<false, false><<<sgrid, 256, 0, stream>>>(STEP_ARGS);
#undef STEP_ARGS
  } else {
  const unsigned pgrid = (unsigned)std::min<uint64_t>((n_seeds + 255) / 256, 256 * 32);
  const PackedIn pk0{ nullptr, 0, 0 };
  uint64_t* key = ctx->w_seed_key.as<uint64_t>();
  static const bool env_explicit_info = getenv("PSIGPU_EXPLICIT_INFO") != nullptr;      // A/B
  uint2* info = (uniform && !env_explicit_info) ? nullptr : ctx->w_seed_info.as<uint2>();       // (equal lengths: derived from the seed's number, run_pipeline)
#define PACK_ARGS(SO) a.d_in, a.d_off, SO, a.nr, d_params, n_seeds, a.nb, k, step, key, info, ctr, nullptr, nullptr, 0
  if (uniform && a.pk) k_seed_pack<false, true, true><<<pgrid, 256, 0, stream>>>(PACK_ARGS(nullptr), *a.pk, un);
  else if (uniform) k_seed_pack<false, false, true><<<pgrid, 256, 0, stream>>>(PACK_ARGS(nullptr), pk0, un);
  else if (a.pk) k_seed_pack<false, true><<<pgrid, 256, 0, stream>>>(PACK_ARGS(ctx->w_seed_off.as<uint64_t>()), *a.pk);
  else k_seed_pack<false><<<pgrid, 256, 0, stream>>>(PACK_ARGS(ctx->w_seed_off.as<uint64_t>()));
#undef PACK_ARGS
  uint64_t n_waves = std::min<uint64_t>(WAVES_MAX, (n_seeds + 15) / 16);
  n_waves = (n_waves + 3) / 4 * 4;
  const uint32_t per_wave = (uint32_t)(((n_seeds + n_waves - 1) / n_waves + 63) / 64 * 64);
  const unsigned grid = (unsigned)(n_waves / 4);
  static const bool env_res16 = getenv("PSIGPU_RES16") != nullptr;
  const bool res8 = !env_res16 && !ctx->opt_res16 && ctx->max_node_len < (1ull << R8_NOFF_BITS);
  uint4* res = ctx->w_seedres.as<uint4>();
  uint64_t* tiles = ctx->w_iv_tiles.as<uint64_t>();
  uint64_t* tiles_off = ctx->w_iv_tiles_off.as<uint64_t>();
  if (res8) k_kmer_probe<true><<<grid, 256, 0, stream>>>(kt, key, d_params, n_seeds, per_wave, ctx->fast_on, ctx->fast_off, thr, res, tiles, tiles_off, ctr);
  else k_kmer_probe<false><<<grid, 256, 0, stream>>>(kt, key, d_params, n_seeds, per_wave, ctx->fast_on, ctx->fast_off, thr, res, tiles, tiles_off, ctr);
  if (res8) k_kmer_emit<true><<<grid, 256, 0, stream>>>(mv, res, ctx->kt_ext.as<KmerSlot>(), oe, tiles, tiles_off, d_params, n_seeds, per_wave, info,
                                                         a.rec_base, d_hits, a.cap, ctr, ctx->fast_on, ctx->fast_off, thr, un.spr, step, env_plain_stores);
  else k_kmer_emit<false><<<grid, 256, 0, stream>>>(mv, res, ctx->kt_ext.as<KmerSlot>(), oe, tiles, tiles_off, d_params, n_seeds, per_wave, info,
                                                    a.rec_base, d_hits, a.cap, ctr, ctx->fast_on, ctx->fast_off, thr, un.spr, step, env_plain_stores);
  }
  if (a.want_sort && a.cap) {
    int fsr = HitSorter::fix_grouped(d_hits, a.cap, &ctr->n_hits_tab.v, (uint64_t*)&ctr->not_grouped.v, stream, &ctx->err);
    if (fsr != PSIGPU_OK) return fsr;
  }
  unsigned long long* h_wflag = reinterpret_cast<unsigned long long*>((char*)fs.h + sizeof(DevCounters) + 32);
  *h_wflag = 0;
  if (a.cap && a.wire) {
    if (a.wf.bytes == 8)
      k_hits_wire8<<<2048, 256, 0, stream>>>(d_hits, &ctr->n_hits_tab.v, &ctr->n_hits_off.v, 0, a.cap, ctx->id_base, a.rec_base, a.wf,
                                             a.wire->as<uint64_t>(), reinterpret_cast<unsigned long long*>((char*)fs.h_dev + sizeof(DevCounters) + 32));
    else
      k_hits_wire16<<<2048, 256, 0, stream>>>(d_hits, &ctr->n_hits_tab.v, &ctr->n_hits_off.v, 0, a.cap, ctx->id_base, a.rec_base, a.wire->as<uint4>());
  }
  k_publish<<<1, 256, 0, stream>>>(reinterpret_cast<const uint4*>(ctr), reinterpret_cast<uint4*>(fs.h_dev), (uint32_t)(sizeof(DevCounters) / 16));
  HIPCHK(ctx, hipEventRecord(fs.done, stream));
  HIPCHK(ctx, hipGetLastError());
  return PSIGPU_OK;
}

// PSIGPU_SORT_UNIQUE on the device: ctx->w_hits[0..n) -> dst[0..*n_out), on `stream`; one host
// synchronisation for the count.  Returns PSIGPU_ERR_FORMAT (err untouched) when the records of this
// chunk do not fit the sorter's 64-bit key: the caller falls back to the host sort.
// Sorted, duplicate-free records of the n hits in w_hits: in `dst`, or -- *in_place -- in w_hits itself when
// ordering the few hits of each seed was all there was to do (HitSorter::fix_grouped).
static int device_sort_unique(psigpu_ctx* ctx, uint64_t n, uint64_t n_reads, uint64_t rec_offset, DevBuf& dst,
                              hipStream_t stream, uint64_t* n_out, bool* in_place)
{
  *n_out = 0;
  *in_place = false;
  if (n == 0) return PSIGPU_OK;
  HIPCHK(ctx, ctx->w_count.ensure(64));
  uint64_t* h_n = reinterpret_cast<uint64_t*>((char*)ctx->h_pinned + sizeof(DevCounters) + 16);
  uint4* h_n_dev = reinterpret_cast<uint4*>((char*)ctx->h_pinned_dev + sizeof(DevCounters) + 16);
  const bool no_grouped = getenv("PSIGPU_NO_GROUPED_SORT") != nullptr;      // tests: always the general path
  if (ctx->grouped_state == 1 && !no_grouped) {                // run_pipeline ordered the groups, and that was all
    ctx->last.n_hits = *n_out = n;
    ctx->last.sorted_in_place = 1;
    *in_place = true;
    return PSIGPU_OK;
  }
  if (ctx->grouped_state == 0 && ctx->last.traverse_launches == 0 && !no_grouped) {       // (the traverser's hits are not grouped by seed)
    EVREC(11, stream);
    int st = HitSorter::fix_grouped(ctx->w_hits.as<psigpu_hit>(), n, nullptr, ctx->w_count.as<uint64_t>(), stream, &ctx->err);
    if (st != PSIGPU_OK) return st;
    EVREC(0, stream);
    k_publish<<<1, 256, 0, stream>>>(ctx->w_count.as<uint4>(), h_n_dev, 1);
    HIPCHK(ctx, hipStreamSynchronize(stream));
    if (*h_n == 0) {
      float t = 0;
      (void)hipEventElapsedTime(&t, ctx->ev[11], ctx->ev[0]);
      ctx->last.ms_sort = t;
      ctx->last.n_hits = *n_out = n;
      ctx->last.sorted_in_place = 1;
      *in_place = true;
      return PSIGPU_OK;
    }
  }
  if (!HitSorter::fits(n_reads, ctx->last_max_read_len, ctx->n_nodes, ctx->max_node_len)) return PSIGPU_ERR_FORMAT;
  HIPCHK(ctx, dst.ensure((n + 1) * sizeof(psigpu_hit)));
  EVREC(11, stream);
  int st = ctx->sorter.run(ctx->w_hits.as<psigpu_hit>(), n, rec_offset, n_reads, ctx->last_max_read_len, ctx->n_nodes,
                           ctx->max_node_len, ctx->id_affine, ctx->id_base, ctx->ids_sorted.as<uint64_t>(),
                           dst.as<psigpu_hit>(), ctx->w_count.as<uint64_t>(), stream, &ctx->err);
  if (st != PSIGPU_OK) return st;
  EVREC(0, stream);                 // (event 0 is free again once run_pipeline has read its times)
  k_publish<<<1, 256, 0, stream>>>(ctx->w_count.as<uint4>(), h_n_dev, 1);
  HIPCHK(ctx, hipStreamSynchronize(stream));
  *n_out = *h_n;
  float t = 0;
  (void)hipEventElapsedTime(&t, ctx->ev[11], ctx->ev[0]);
  ctx->last.ms_sort = t;
  ctx->last.n_hits = *n_out;
  return PSIGPU_OK;
}

static int find_seeds_device(psigpu_ctx* ctx, const char* d_bases, const PackedIn* packed, const uint64_t* d_read_off,
                             uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step,
                             uint64_t rec_offset, uint32_t flags, void* stream,
                             const psigpu_hit** d_hits, uint64_t* n_hits)
{
  if (!ctx || !d_hits || !n_hits || (n_reads && (!d_read_off))) return PSIGPU_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if ((flags & PSIGPU_ALL) == 0) flags |= PSIGPU_ALL;
  uint64_t n = 0;
  int st = run_pipeline(ctx, d_bases, d_read_off, n_reads, n_bases, k, step, rec_offset, flags, (hipStream_t)stream, &n, nullptr, packed);
  if (st != PSIGPU_OK) return st;
  *d_hits = ctx->w_hits.as<psigpu_hit>();
  *n_hits = n;
  if (flags & PSIGPU_SORT_UNIQUE) {
    bool in_place = false;
    st = device_sort_unique(ctx, n, n_reads, rec_offset, ctx->w_sorted[0], (hipStream_t)stream, &n, &in_place);
    if (st == PSIGPU_ERR_FORMAT) { ctx->err = "hit records of this chunk do not fit the device sorter's 64-bit key"; return PSIGPU_ERR_ARG; }
    if (st != PSIGPU_OK) return st;
    *d_hits = (n && !in_place) ? ctx->w_sorted[0].as<psigpu_hit>() : ctx->w_hits.as<psigpu_hit>();
    *n_hits = n;
  }
  return PSIGPU_OK;
}

int psigpu_find_seeds_device(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_read_off,
                             uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step,
                             uint64_t rec_offset, uint32_t flags, void* stream,
                             const psigpu_hit** d_hits, uint64_t* n_hits)
{
  if (ctx && ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  return find_seeds_device(ctx, d_bases, nullptr, d_read_off, n_reads, n_bases, k, step, rec_offset, flags, stream, d_hits, n_hits);
}

int psigpu_find_seeds_device_packed(psigpu_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_n_mask,
                                    const uint64_t* d_read_off, uint64_t n_reads, uint64_t n_bases, uint32_t k,
                                    uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream,
                                    const psigpu_hit** d_hits, uint64_t* n_hits)
{
  if (n_bases && !d_packed) return PSIGPU_ERR_ARG;
  if (ctx && ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  const PackedIn pk{ d_n_mask, 0, 0 };
  return find_seeds_device(ctx, reinterpret_cast<const char*>(d_packed), &pk, d_read_off, n_reads, n_bases, k, step, rec_offset, flags,
                           stream, d_hits, n_hits);
}

// ---- two chunks in flight through the device-resident entry ----------------------------------------------------
// A step of the default mode is ~0.35 ms of kernels and ~0.04 ms in which the device waits for the host: the
// synchronisation, the counters read, the return to the caller, the caller's next call, five launches.  A caller that
// has the next chunk ready (a pipeline that double-buffers its read batches, as psikt's host entry does internally with
// its sub-batches) begins it before it ends the current one: _begin queues a chunk's kernels and returns, _end waits
// for the OLDEST chunk begun and hands out its hits.  At most two chunks are begun at a time (on ONE stream); the hits of a
// chunk stay valid until the next _end.  A chunk that needs more than the default mode's kernels (another query
// mode, a traverser pass, tables not made yet, buffers that would have to grow under a chunk in flight) is answered by the
// synchronous entry inside its _end -- same result, no overlap.
static int dev_begin(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_mask, bool packed, const uint64_t* d_read_off,
                     uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream)
{
  if (!ctx || (n_reads && !d_read_off) || (packed && n_bases && !d_bases)) return PSIGPU_ERR_ARG;
  if (ctx->dp_count == 2) { ctx->err = "two chunks are begun already: end one first"; return PSIGPU_ERR_STATE; }
  // the chunks in flight share one workspace (seed keys, probe results, counters) and only stream order keeps them apart
  if (ctx->dp_count && stream != ctx->dp_stream) {
    ctx->err = "a chunk is in flight on another stream: every chunk begun before the oldest is ended must use the same stream";
    return PSIGPU_ERR_STATE;
  }
  ctx->dp_stream = stream;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if ((flags & PSIGPU_ALL) == 0) flags |= PSIGPU_ALL;
  psigpu_ctx::DevPending& p = ctx->dpend[(ctx->dp_head + ctx->dp_count) % 2];
  p = psigpu_ctx::DevPending{};
  p.d_bases = d_bases; p.d_mask = d_mask; p.packed = packed; p.d_off = d_read_off; p.nr = n_reads; p.nb = n_bases;
  p.rec_offset = rec_offset; p.k = k; p.step = step; p.flags = flags; p.stream = stream;
  const uint32_t stp = step ? step : k;
  static const bool env_no_lookahead = getenv("PSIGPU_NO_LOOKAHEAD") != nullptr;
  const bool want_sort = (flags & PSIGPU_SORT_UNIQUE) != 0;
  bool fast = n_reads && n_bases && k && ctx->fast_k == k && ctx->fast_flags == (flags & PSIGPU_ALL) && ctx->lkt_ready && ctx->kt_ready &&
              ctx->lkt_k == k && !env_no_lookahead && !ctx->opt_no_lookahead && !getenv("PSIGPU_TRACE") &&
              !(want_sort && getenv("PSIGPU_NO_GROUPED_SORT") != nullptr);
  const uint64_t seeds_max = n_bases / stp + n_reads;
  if (fast && seeds_max >= 0xFFFFFFF0ull) fast = false;
  if (fast) {
    const uint64_t cap = std::max<uint64_t>(ctx->hits_cap_hint, 2 * seeds_max + (1u << 16));
    const int q = (int)(ctx->dp_seq % psigpu_ctx::N_FAST);
    psigpu_ctx::FastSlot& fs = ctx->fast[q];
    // nothing may be regrown (freed) under a chunk in flight: with one begun, a buffer that is too small makes this chunk a synchronous one
    struct Need { DevBuf* b; size_t bytes; };
    const bool fz = fused_step_wanted(ctx);        // (one kernel per chunk: no key / result arrays)
    const Need needs[] = { { &ctx->w_ctr, sizeof(DevCounters) }, { &ctx->w_total, 64 }, { &ctx->w_tiles, (n_reads / SCAN_TILE + 2) * 8 },
                           { &ctx->w_seed_off, (n_reads + 1) * 8 }, { &ctx->w_seed_key, fz ? 0 : (seeds_max + 1) * 8 },
                           { &ctx->w_seed_info, fz ? 0 : (seeds_max + 1) * 8 }, { &ctx->w_seedres, fz ? 0 : (seeds_max + 16) * 16 },
                           { &ctx->w_iv_tiles, fz ? (size_t)0 : (size_t)(WAVES_MAX + 8) * 8 }, { &ctx->w_iv_tiles_off, fz ? (size_t)0 : (size_t)(WAVES_MAX + 8) * 8 },
                           { &ctx->w_tilestate, fz ? (seeds_max / KS_TILE + 66) * 8 : 0 },
                           { &fs.hits, (cap + 1) * sizeof(psigpu_hit) } };
    bool grow = !fs.h;
    for (const Need& nd : needs) grow = grow || nd.b->cap < nd.bytes;
    if (grow && ctx->dp_count) fast = false;
    else if (grow) {
      // (all three slots at once: the next chunk is begun with this one in flight and must find its slot made)
      hipError_t e = hipSuccess;
      for (const Need& nd : needs) if (e == hipSuccess && nd.b != &fs.hits) e = nd.b->ensure(nd.bytes);
      for (int s3 = 0; s3 < psigpu_ctx::N_FAST; ++s3) {
        psigpu_ctx::FastSlot& f3 = ctx->fast[s3];
        if (f3.hits.p && f3.hits.cap < (cap + 1) * sizeof(psigpu_hit)) {      // (the last _end may have handed this buffer out)
          ctx->retired.push_back(f3.hits.p);
          f3.hits.p = nullptr; f3.hits.cap = 0;
        }
        if (e == hipSuccess) e = f3.hits.ensure((cap + 1) * sizeof(psigpu_hit));
        if (e == hipSuccess && !f3.h) {
          e = hipHostMalloc(&f3.h, sizeof(DevCounters) + 64, hipHostMallocMapped);
          if (e == hipSuccess) e = hipHostGetDevicePointer(&f3.h_dev, f3.h, 0);
          if (e == hipSuccess) e = hipEventCreate(&f3.begin);
          if (e == hipSuccess) e = hipEventCreate(&f3.done);
        }
      }
      if (e != hipSuccess) { (void)hipGetLastError(); fast = false; }       // (the synchronous entry reports what is wrong, if anything is)
    }
    if (fast) {
      PackedIn pk{ d_mask, 0, 0 };
      FastArgs a;
      a.d_in = d_bases; a.pk = packed ? &pk : nullptr; a.d_off = d_read_off;
      a.nr = n_reads; a.nb = n_bases; a.k = k; a.step = stp; a.rec_base = rec_offset;
      a.want_sort = want_sort; a.claim_uniform = (flags & PSIGPU_UNIFORM_READS) != 0;
      a.wf = WireFmt{}; a.wire = nullptr; a.cap = cap; a.slot = q;
      p.cap = cap; p.slot = q;
      int st = enqueue_default(ctx, a, (hipStream_t)stream, &p.serial, &p.uniform);
      if (st != PSIGPU_OK) return st;
      p.queued = true;
      ++ctx->dp_seq;
    }
  }
  ++ctx->dp_count;
  return PSIGPU_OK;
}

int psigpu_find_seeds_device_begin(psigpu_ctx* ctx, const char* d_bases, const uint64_t* d_read_off, uint64_t n_reads, uint64_t n_bases,
                                   uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, void* stream)
{
  return dev_begin(ctx, d_bases, nullptr, false, d_read_off, n_reads, n_bases, k, step, rec_offset, flags, stream);
}

int psigpu_find_seeds_device_packed_begin(psigpu_ctx* ctx, const uint64_t* d_packed, const uint64_t* d_n_mask, const uint64_t* d_read_off,
                                          uint64_t n_reads, uint64_t n_bases, uint32_t k, uint32_t step, uint64_t rec_offset,
                                          uint32_t flags, void* stream)
{
  return dev_begin(ctx, reinterpret_cast<const char*>(d_packed), d_n_mask, true, d_read_off, n_reads, n_bases, k, step, rec_offset, flags, stream);
}

int psigpu_find_seeds_device_end(psigpu_ctx* ctx, const psigpu_hit** d_hits, uint64_t* n_hits)
{
  if (!ctx || !d_hits || !n_hits) return PSIGPU_ERR_ARG;
  if (ctx->dp_count == 0) { ctx->err = "no chunk was begun"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (void* old : ctx->retired) (void)hipFree(old);       // (what the previous _end handed out ends its life here: psi_gpu.h)
  ctx->retired.clear();
  const psigpu_ctx::DevPending p = ctx->dpend[ctx->dp_head];
  ctx->dp_head = (ctx->dp_head + 1) % 2;
  --ctx->dp_count;
  if (p.queued) {
    psigpu_ctx::FastSlot& fs = ctx->fast[p.slot];
    {
      // (polled, not hipEventSynchronize: with the next chunk queued behind the event the runtime would wait for that too)
      hipError_t qe;
      uint32_t spins = 0;
      while ((qe = hipEventQuery(fs.done)) == hipErrorNotReady) { if (++spins > 2000) std::this_thread::yield(); }
      if (qe != hipSuccess) { ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(qe); return PSIGPU_ERR_DEVICE; }
    }
    const DevCounters& h = *reinterpret_cast<const DevCounters*>(fs.h);
    const uint64_t n = h.n_hits_tab.v;
    const bool want_sort = (p.flags & PSIGPU_SORT_UNIQUE) != 0;
    if (h.serial.v != p.serial) ++ctx->stale_handbacks;
    if (h.serial.v == p.serial && !(p.uniform && h.not_uniform.v) && n <= p.cap && !(want_sort && h.not_grouped.v)) {
      psigpu_counters pc{};
      pc.n_reads = p.nr; pc.n_seeds = h.n_seeds_true.v; pc.n_seeds_valid = h.n_seeds_valid.total();
      pc.n_seeds_on_path = h.n_live.total(); pc.n_hits_on_path = h.hits_on(); pc.n_hits_off_path = n - h.hits_on(); pc.n_hits = n;
      pc.n_loci = ctx->n_loci; pc.n_locus_kmers = ctx->fast_off ? ctx->lkt_n_ent : 0; pc.n_path_kmers = ctx->kt_n_path_kmers;
      pc.ms_locus_table_build = ctx->lkt_build_ms;
      { float t = 0; (void)hipEventElapsedTime(&t, fs.begin, fs.done); pc.ms_total = t; }
      pc.sorted_in_place = want_sort ? 1u : 0u;
      pc.lookahead_subbatches = 1;
      pc.fused_step = fused_step_wanted(ctx) ? 1u : 0u;
      pc.stale_handbacks = ctx->stale_handbacks;
      ctx->last = pc;
      ctx->hits_cap_hint = std::max<uint64_t>(ctx->hits_cap_hint, n + n / 8);
      *d_hits = fs.hits.as<psigpu_hit>();
      *n_hits = n;
      return PSIGPU_OK;
    }
    // something the five kernels alone do not settle (reads of several lengths behind the claim, more hits than the
    // slot holds, records not grouped by seed): the chunk again, the synchronous way, behind whatever else is queued
    ++ctx->lookahead_fallbacks;
  }
  const PackedIn pk{ p.d_mask, 0, 0 };
  return find_seeds_device(ctx, p.d_bases, p.packed ? &pk : nullptr, p.d_off, p.nr, p.nb, p.k, p.step, p.rec_offset, p.flags, p.stream, d_hits, n_hits);
}

int psigpu_find_mems(psigpu_ctx* ctx, const char* bases, const uint64_t* read_off, uint64_t n_reads, uint32_t minlen,
                     uint32_t max_mem, uint64_t rec_offset, psigpu_mems* out)
{
  if (!ctx || !out || (n_reads && (!read_off || !bases))) return PSIGPU_ERR_ARG;
  out->n = 0; out->data = nullptr;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!ctx->have_graph || !ctx->have_index) { ctx->err = "graph / index not loaded"; return PSIGPU_ERR_STATE; }
  if (minlen == 0) { ctx->err = "minimum match length must be positive"; return PSIGPU_ERR_ARG; }
  if (n_reads >= 0xFFFFFFF0ull) { ctx->err = "too many reads in one chunk"; return PSIGPU_ERR_ARG; }
  if (n_reads == 0 || ctx->n_paths == 0) return PSIGPU_OK;        // length( indexText ) == 0: nothing on paths (:1467)
  if (ctx->sa_rate != 1 || !ctx->p0().have_text4) {
    ctx->err = "MEM mode needs the whole suffix array and the text on the device (sa_rate 1)";
    return PSIGPU_ERR_STATE;
  }
  const uint64_t n_bases = read_off[n_reads];
  for (uint64_t r = 0; r < n_reads; ++r)
    if (read_off[r + 1] - read_off[r] >= 0xFFFFFFF0ull) { ctx->err = "read too long"; return PSIGPU_ERR_ARG; }
  HIPCHK(ctx, ctx->w_bases.ensure(n_bases + 64));
  HIPCHK(ctx, ctx->w_read_off.ensure((n_reads + 1) * 8));
  if (n_bases) HIPCHK(ctx, hipMemcpy(ctx->w_bases.p, bases, n_bases, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->w_read_off.p, read_off, (n_reads + 1) * 8, hipMemcpyHostToDevice));
  TmpBuf groups, ctr, cnt, goff, tiles, hits;
  HIPCHK(ctx, ctr.alloc(64));
  const uint32_t thr = ctx->gocc_thr ? ctx->gocc_thr : 0xFFFFFFFFu;
  const uint32_t mm = max_mem ? max_mem : 0xFFFFFFFFu;
  MemParts mp{};
  mp.n_parts = (uint32_t)ctx->parts.size();
  for (uint32_t q = 0; q < mp.n_parts; ++q) {
    const psigpu_ctx::FmPart& fp = *ctx->parts[q];
    mp.p[q] = MemPart{ fp.samples.as<uint32_t>(), fp.text4.as<uint64_t>(), fp.seg.as<SegRec>(), fp.seg_dir.as<uint32_t>(), (uint32_t)fp.text_len };
  }
  // a reported pattern consumes at least minlen + 1 bases and leaves one group per part it occurs in
  uint64_t cap_groups = (n_bases / std::max<uint32_t>(1, minlen) + n_reads) * mp.n_parts + 1024;
  unsigned long long h[2] = { 0, 0 };
  for (int attempt = 0; attempt < 2; ++attempt) {
    HIPCHK(ctx, groups.alloc(cap_groups * sizeof(MemGroup)));
    HIPCHK(ctx, hipMemset(ctr.p, 0, 64));
    k_find_mems<<<(unsigned)((n_reads + 63) / 64), 64>>>(ctx->w_bases.as<char>(), ctx->w_read_off.as<uint64_t>(), n_reads,
                                                         mp, minlen, thr, mm, groups.as<MemGroup>(), cap_groups,
                                                         ctr.as<unsigned long long>(), ctr.as<unsigned long long>() + 1);
    HIPCHK(ctx, hipMemcpy(h, ctr.p, 16, hipMemcpyDeviceToHost));
    if (h[0] <= cap_groups) break;
    if (attempt) { ctx->err = "MEM group buffer overflow"; return PSIGPU_ERR_NOMEM; }
    cap_groups = h[0] + 16;
  }
  const uint64_t n_groups = h[0], n_hits = h[1];
  if (n_hits == 0) return PSIGPU_OK;
  // group sizes -> offsets, then one record per occurrence
  const uint64_t n_tiles = n_groups / SCAN_TILE + 1;
  HIPCHK(ctx, cnt.alloc((n_groups + 1) * 4));
  HIPCHK(ctx, goff.alloc((n_groups + 2) * 8));
  HIPCHK(ctx, tiles.alloc(n_tiles * 8));
  HIPCHK(ctx, hits.alloc(n_hits * sizeof(MemHit)));
  k_mem_counts<<<(unsigned)((n_groups + 255) / 256), 256>>>(groups.as<MemGroup>(), n_groups, cnt.as<uint32_t>());
  k_scan_tiles<<<(unsigned)n_tiles, SCAN_THREADS>>>(cnt.as<uint32_t>(), n_groups, tiles.as<uint64_t>());
  k_scan_sums<<<1, SCAN_THREADS>>>(tiles.as<uint64_t>(), n_tiles, ctr.as<uint64_t>() + 2);
  k_scan_final<<<(unsigned)n_tiles, SCAN_THREADS>>>(cnt.as<uint32_t>(), n_groups, tiles.as<uint64_t>(), goff.as<uint64_t>());
  k_mem_locate<<<(unsigned)((n_groups + 255) / 256), 256>>>(groups.as<MemGroup>(), goff.as<uint64_t>(), n_groups, mp, rec_offset,
                                                            hits.as<MemHit>());
  psigpu_mem_hit* hp = (psigpu_mem_hit*)g_pinned.get(n_hits * sizeof(psigpu_mem_hit));
  if (!hp) { ctx->err = "cannot allocate pinned host memory for the hits"; return PSIGPU_ERR_NOMEM; }
  hipError_t e = hipMemcpy(hp, hits.p, n_hits * sizeof(psigpu_mem_hit), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { g_pinned.put(hp); ctx->err = hipGetErrorString(e); return PSIGPU_ERR_DEVICE; }
  // the order groups were appended in depends on scheduling: hand the records out in the order the
  // reference's loop produces them per read -- (read, read offset) -- and by position inside a group
  std::sort(hp, hp + n_hits, [](const psigpu_mem_hit& a, const psigpu_mem_hit& b) {
    if (a.read_id != b.read_id) return a.read_id < b.read_id;
    if (a.read_offset != b.read_offset) return a.read_offset < b.read_offset;
    if (a.node_id != b.node_id) return a.node_id < b.node_id;
    return a.node_offset < b.node_offset;
  });
  out->data = hp;
  out->n = n_hits;
  return PSIGPU_OK;
}

void psigpu_free_mems(psigpu_mems* mems)
{
  if (!mems) return;
  if (mems->data) g_pinned.put(mems->data);
  mems->data = nullptr;
  mems->n = 0;
}

int psigpu_copy_hits(psigpu_ctx* ctx, psigpu_hit* host_dst, const psigpu_hit* d_src, uint64_t n)
{
  if (!ctx || (n && (!host_dst || !d_src))) return PSIGPU_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n) HIPCHK(ctx, hipMemcpy(host_dst, d_src, n * sizeof(psigpu_hit), hipMemcpyDeviceToHost));
  return PSIGPU_OK;
}

int psigpu_prepare(psigpu_ctx* ctx, uint32_t k)
{
  if (!ctx) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (k == 0 || k > PSIGPU_MAX_SEED_LEN) { ctx->err = "seed length out of range (1..63)"; return PSIGPU_ERR_ARG; }
  if (!ctx->have_graph || !ctx->have_index) { ctx->err = "graph / index not loaded"; return PSIGPU_ERR_STATE; }
  resolve_auto_mode(ctx);
  if (ctx->index_k != k || k > PSIGPU_MAX_TABLE_SEED_LEN) return PSIGPU_OK;   // tables exist for the index's seed length only, and for one-word seeds
  int st = PSIGPU_OK;
  if (ctx->query_mode == PSIGPU_MODE_TRAVERSE) {                           // (only the paths' k-mers are tabulated in traverse mode)
    if (ctx->n_paths && ctx->sa_rate == 1 && !(ctx->tune & PSIGPU_TUNE_NO_PATH_TABLE)) st = ensure_lkt(ctx, k, graph_view(ctx));
    // ... and the loci's prefix walks, where the traverser starts
    if (st == PSIGPU_OK && ctx->n_loci && k > PFX_SHORT && !ctx->opt_no_pfx_roots && getenv("PSIGPU_NO_PFX_ROOTS") == nullptr)
      st = ensure_pfx_roots(ctx, graph_view(ctx));
  } else if (!(ctx->n_loci == 0 && !(ctx->query_mode == PSIGPU_MODE_KMER_TABLE && ctx->n_paths)))
    st = ensure_lkt(ctx, k, graph_view(ctx));
  // the FM modes' per-row records (the k-mer table mode never reads them)
  if (st == PSIGPU_OK && !(ctx->lkt_ready && ctx->kt_ready) && ctx->n_paths && ctx->fm_ok && !ctx->rows_tried) {
    st = build_row_records(ctx, ctx->index_k);
    ctx->rows_tried = st == PSIGPU_OK;
  }
  return st == PSIGPU_OK ? loader_fence(ctx) : st;
}

void* psigpu_host_alloc(uint64_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void psigpu_host_free(void* p)
{
  if (p) (void)hipHostFree(p);
}

int psigpu_measure_random_loads(psigpu_ctx* ctx, uint64_t table_bytes, uint64_t n_loads, uint32_t quad_sectors, double* loads_per_s)
{
  if (!ctx || !loads_per_s || table_bytes < (1u << 20)) return PSIGPU_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  TmpBuf t, out;
  if (t.alloc(table_bytes) != hipSuccess) { (void)hipGetLastError(); ctx->err = "not enough device memory for the measurement table"; return PSIGPU_ERR_NOMEM; }
  HIPCHK(ctx, out.alloc(64));
  HIPCHK(ctx, hipMemset(t.p, 1, table_bytes));
  const uint32_t blocks = 8192;                                   // every wave slot of the chip, as the query kernels are launched
  const uint64_t threads = (uint64_t)blocks * 256;
  const uint64_t per_thread = quad_sectors ? 4 * n_loads : n_loads;  // a quad's four lanes make ONE sector load
  const uint32_t iters = (uint32_t)std::max<uint64_t>(1, (per_thread + threads - 1) / threads);
  hipEvent_t a, b;
  HIPCHK(ctx, hipEventCreate(&a)); HIPCHK(ctx, hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {                             // (the first one pays the page-table walk)
    (void)hipEventRecord(a, nullptr);
    if (quad_sectors) k_rand_loads<true><<<blocks, 256>>>(t.as<uint4>(), table_bytes / 64, iters, out.as<uint32_t>());
    else k_rand_loads<false><<<blocks, 256>>>(t.as<uint4>(), table_bytes / 64, iters, out.as<uint32_t>());
    (void)hipEventRecord(b, nullptr);
    if (hipEventSynchronize(b) != hipSuccess) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); ctx->err = "measurement kernel failed"; return PSIGPU_ERR_DEVICE; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep && ms < best) best = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  const double done = (double)threads * iters / (quad_sectors ? 4.0 : 1.0);
  *loads_per_s = done / (best * 1e-3);
  return PSIGPU_OK;
}

}  // extern "C"

namespace {

// is this host pointer pinned (hipHostMalloc / hipHostRegister), i.e. can the copy engine read it in place?
// `delta` = what to add to the host address to get the address the device side uses for the same
// byte (0 for hipHostMalloc memory; memory pinned later with hipHostRegister may be mapped elsewhere)
bool host_ptr_is_pinned(const void* p, ptrdiff_t* delta)
{
  *delta = 0;
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeHost) return false;
  if (a.devicePointer && a.hostPointer) *delta = (const char*)a.devicePointer - (const char*)a.hostPointer;
  return true;
}

// pageable -> pinned staging copy with a few threads (one core moves ~10 GB/s, the link 55)
void parallel_copy(char* dst, const char* src, size_t n)
{
  const size_t MIN_PART = 2u << 20;
  unsigned parts = (unsigned)std::min<size_t>(4, std::max<size_t>(1, n / MIN_PART));
  if (parts <= 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n / parts + 63) & ~(size_t)63;
  for (unsigned t = 1; t < parts; ++t) {
    const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per);
    th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
  }
  memcpy(dst, src, std::min(n, per));
  for (auto& t : th) t.join();
}

}  // namespace

namespace {

// Widening of the wire records (k_hits_wire16) into the caller's 32-byte records on a few host threads, sub-batch
// after sub-batch, while the pipeline goes on.  Job j: `n` records from a slot's pinned landing buffer to `dst`;
// thread 0 waits for the slot's transfer, then every thread widens its slice.
// one 32-byte record into the caller's (pinned) array with two streaming stores: the array is written once, front
// to back, 224 MB per 1 M-read chunk -- ordinary stores would first READ every line they fill
static inline void store_hit(psigpu_hit* dst, uint64_t node, uint64_t noff, uint64_t read, uint64_t roff)
{
  typedef long long v2 __attribute__((vector_size(16)));
  v2 lo = { (long long)node, (long long)noff }, hi = { (long long)read, (long long)roff };
  __builtin_nontemporal_store(lo, reinterpret_cast<v2*>(dst));
  __builtin_nontemporal_store(hi, reinterpret_cast<v2*>(dst) + 1);
}

// One helper thread that lives with the context and runs one job at a time (the stager of pageable reads: a call used to
// start and join a thread of its own).
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<void()> job;
  bool has = false, busy = false, quit = false;
  void run(std::function<void()> f)
  {
    { std::lock_guard<std::mutex> lk(mu); job = std::move(f); has = true; busy = true; }
    if (!th.joinable()) th = std::thread([this] { loop(); });
    cv.notify_one();
  }
  void wait()                                   // the job, if any, has returned
  {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return !busy; });
  }
  void loop()
  {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || has; });
        if (quit) return;
        f = std::move(job); has = false;
      }
      f();
      { std::lock_guard<std::mutex> lk(mu); busy = false; }
      cv_done.notify_all();
    }
  }
  ~Worker()
  {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

struct Widener {
  struct Job { const void* src; psigpu_hit* dst; uint64_t n, id_base, rec_base; int slot; WireFmt fmt; };
  // The threads live as long as the context (round 4: a call used to start and join up to eight threads of its own,
  // 0.15-0.2 ms of a 2-ms call); a call is a SESSION: begin() resets the job list and wakes them, end() waits until
  // every one of them has left the session.
  std::vector<Job> jobs;
  size_t n_jobs = 0;
  std::atomic<size_t> posted{ 0 }, ready{ 0 };
  // slices done, PER JOB: thread 0 may be a job ahead of a thread that was descheduled inside the job before, so a
  // count over all jobs reaches "T x (j + 1)" while a slice of job j is still being read (seen under three fuzz
  // processes on one box: a record of the sub-batch that reused the landing buffer)
  std::unique_ptr<std::atomic<uint32_t>[]> parts;
  size_t parts_cap = 0;
  size_t checked = 0;                           // caller's thread only: jobs [0, checked) are known to be finished
  std::atomic<bool> stop{ false };              // the session is abandoned (error path): leave it
  std::vector<std::thread> th;
  std::function<void(int)> wait_copy;           // blocks until the slot's device-to-host transfer is complete
  unsigned T = 0;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t session = 0;                         // (mu)
  bool quit = false;                            // (mu)
  std::atomic<unsigned> left{ 0 };              // threads that have left the current session
  bool open = false;                            // caller's thread only: a session is running

  void begin(unsigned n_threads, size_t n_jobs_, std::function<void(int)> wc)
  {
    if (T == 0) {
      T = n_threads;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([this, t] { loop(t); });
    }
    n_jobs = n_jobs_;
    if (jobs.size() < n_jobs) jobs.resize(n_jobs);
    if (parts_cap < n_jobs) { parts_cap = n_jobs + n_jobs / 2 + 16; parts.reset(new std::atomic<uint32_t>[parts_cap]); }
    for (size_t j = 0; j < n_jobs; ++j) parts[j].store(0, std::memory_order_relaxed);
    posted.store(0); ready.store(0); stop.store(false); left.store(0);
    checked = 0;
    wait_copy = std::move(wc);
    { std::lock_guard<std::mutex> lk(mu); ++session; }
    cv.notify_all();
    open = true;
  }
  // every thread out of the session (all jobs done, or `stop` after an error): nothing of the call's buffers is touched after this
  void end()
  {
    if (!open) return;
    stop.store(true);
    while (left.load(std::memory_order_acquire) < T) std::this_thread::yield();
    open = false;
  }
  void post(size_t j, const Job& job) { jobs[j] = job; posted.store(j + 1, std::memory_order_release); }
  // every slice of jobs [0, upto) has been widened (called by the thread that posts)
  void wait_finished(size_t upto)
  {
    for (; checked < upto; ++checked)
      while (parts[checked].load(std::memory_order_acquire) < T) std::this_thread::yield();
  }
  void loop(unsigned t)
  {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || session != seen; });
        if (quit) return;
        seen = session;
      }
      run(t);
      left.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void run(unsigned t)
  {
    for (size_t j = 0; j < n_jobs; ++j) {
      while (posted.load(std::memory_order_acquire) <= j) { if (stop.load()) return; std::this_thread::yield(); }
      const Job job = jobs[j];
      if (t == 0) { if (job.n) wait_copy(job.slot); ready.store(j + 1, std::memory_order_release); }
      else while (ready.load(std::memory_order_acquire) <= j) { if (stop.load()) return; std::this_thread::yield(); }
      const uint64_t a = job.n * t / T, b = job.n * (t + 1) / T;
      if (job.fmt.bytes == 8) {
        const uint64_t* src = static_cast<const uint64_t*>(job.src);
        const uint32_t nb = job.fmt.noff_bits, vb = job.fmt.node_bits, rb = job.fmt.roff_bits;
        const uint64_t nm = (1ull << nb) - 1, vm = (1ull << vb) - 1, rm = (1ull << rb) - 1;
        for (uint64_t i = a; i < b; ++i) {
          uint64_t key = src[i];
          const uint64_t noff = key & nm; key >>= nb;
          const uint64_t node = key & vm; key >>= vb;
          const uint64_t roff = key & rm; key >>= rb;
          store_hit(job.dst + i, job.id_base + node, noff, job.rec_base + key, roff);
        }
      } else {
        const uint4* src = static_cast<const uint4*>(job.src);
        for (uint64_t i = a; i < b; ++i) {
          const uint4 w = src[i];
          store_hit(job.dst + i, job.id_base + w.x, w.y, job.rec_base + w.z, w.w);
        }
      }
      __builtin_ia32_sfence();                   // the streaming stores above, before the slice is declared done
      parts[j].fetch_add(1, std::memory_order_acq_rel);
    }
  }
  ~Widener()
  {
    end();
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    for (auto& x : th) if (x.joinable()) x.join();
  }
};

}  // namespace

extern "C" {

static void widener_destroy(psigpu_ctx* ctx)
{
  delete static_cast<Widener*>(ctx->widener);
  ctx->widener = nullptr;
  delete static_cast<Worker*>(ctx->stager);
  ctx->stager = nullptr;
}

// Streams and events of the host entry's pipeline, made on its first call.
//
// The two transfer directions get one SDMA engine each, by name.  hipMemcpyAsync leaves the choice
// to the runtime ("the engine this stream used last if it is idle, else the first idle one"), and
// in this pipeline both directions regularly end up on ONE engine and share its ~55 GB/s instead of
// running full duplex (2 x 47 GB/s, tools/pcie_rate.hip) -- a chunk then costs 7 ms instead of 5.5.
// ROCr's hsa_amd_memory_async_copy_on_engine takes the engine explicitly; completion is an HSA
// signal the host waits on (the pipeline is host-synchronous at every stage boundary anyway).
// When anything here is unavailable the transfers fall back to hipMemcpyAsync.
static hsa_status_t find_agents_cb(hsa_agent_t a, void* data)
{
  auto* v = static_cast<std::vector<hsa_agent_t>*>(data);
  v->push_back(a);
  return HSA_STATUS_SUCCESS;
}

static void engine_copy_init(psigpu_ctx* ctx)
{
  psigpu_ctx::EngineCopy& ec = ctx->ec;
  ec.ok = false;
  if (getenv("PSIGPU_NO_ENGINE_COPY") || ctx->opt_no_engine_copy) return;
  if (!g_hsa.init()) return;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return;
  std::vector<hsa_agent_t> agents;
  if (hsa_iterate_agents(find_agents_cb, &agents) != HSA_STATUS_SUCCESS) return;
  bool have_gpu = false, have_cpu = false;
  for (hsa_agent_t a : agents) {
    hsa_device_type_t type;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) continue;
    if (type == HSA_DEVICE_TYPE_CPU && !have_cpu) { ec.cpu = a; have_cpu = true; }
    if (type == HSA_DEVICE_TYPE_GPU && !have_gpu) {
      uint32_t bdf = 0, domain = 0;
      if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) continue;
      (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
      const uint32_t want = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
      if ((bdf & ~7u) == want && domain == (uint32_t)prop.pciDomainID) { ec.gpu = a; have_gpu = true; }
    }
  }
  if (!have_gpu || !have_cpu) return;
  uint32_t mask_in = 0, mask_out = 0;
  if (hsa_amd_memory_copy_engine_status(ec.gpu, ec.cpu, &mask_in) != HSA_STATUS_SUCCESS) mask_in = 0;      // dst, src
  if (hsa_amd_memory_copy_engine_status(ec.cpu, ec.gpu, &mask_out) != HSA_STATUS_SUCCESS) mask_out = 0;
  if (mask_in == 0 || mask_out == 0) return;
  ec.eng_in = mask_in & (~mask_in + 1);                       // lowest engine that can copy host -> device
  uint32_t rest = mask_out & ~ec.eng_in;
  if (rest == 0) return;                                      // a single engine: nothing to separate
  ec.eng_out = rest & (~rest + 1);
  constexpr int R = psigpu_ctx::EngineCopy::IN_RING;
  for (int i = 0; i < R + 5; ++i) {
    if (!g_hsa.take(i < R ? &ec.sig_in[i] : i < R + 2 ? &ec.sig_out[i - R] : &ec.sig_fast[i - R - 2])) return;
    ec.n_sig = i + 1;
  }
  ec.ok = true;
  if (getenv("PSIGPU_TRACE")) fprintf(stderr, "[psigpu] copy engines: in 0x%x of 0x%x, out 0x%x of 0x%x\n", ec.eng_in, mask_in, ec.eng_out, mask_out);
}

static int pipeline_init(psigpu_ctx* ctx)
{
  if (ctx->s_comp) return PSIGPU_OK;
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_comp, hipStreamNonBlocking));
  engine_copy_init(ctx);
  if (!ctx->ec.ok) {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
  }
  for (auto& sl : ctx->slot) {
    HIPCHK(ctx, hipEventCreateWithFlags(&sl.in_ready, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&sl.out_done, hipEventDisableTiming));
  }
  return PSIGPU_OK;
}

// one transfer of the pipeline: `slot_sig` goes 1 -> 0 when it is complete
// (`n_transfers`: how many transfers will complete on this signal -- packed reads move two arrays per sub-batch;
// every completed transfer takes one off, the waiter waits for 0)
static bool engine_copy_more(psigpu_ctx* ctx, bool to_device, void* dst, const void* src, size_t bytes, hsa_signal_t slot_sig)
{
  const psigpu_ctx::EngineCopy& ec = ctx->ec;
  return hsa_amd_memory_async_copy_on_engine(dst, to_device ? ec.gpu : ec.cpu, src, to_device ? ec.cpu : ec.gpu, bytes,
                                             0, nullptr, slot_sig,
                                             (hsa_amd_sdma_engine_id_t)(to_device ? ec.eng_in : ec.eng_out), true) == HSA_STATUS_SUCCESS;
}

static bool engine_copy(psigpu_ctx* ctx, bool to_device, void* dst, const void* src, size_t bytes, hsa_signal_t slot_sig, int n_transfers = 1)
{
  hsa_signal_store_relaxed(slot_sig, n_transfers);
  if (!engine_copy_more(ctx, to_device, dst, src, bytes, slot_sig)) { hsa_signal_store_relaxed(slot_sig, 0); return false; }
  return true;
}

static void engine_wait_value(hsa_signal_t sig, int64_t below)
{
  while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, below, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= below) { }
}

static void engine_wait(hsa_signal_t sig) { engine_wait_value(sig, 1); }

// One chunk through the host entry point = SURVEY 8(d)'s timed region: H2D of the reads, kernels, D2H
// of the hits.  The chunk is cut into sub-batches of contiguous reads that flow through two slots:
// while sub-batch i is on the compute stream, the copy engines bring in i + 1 and take out i - 1
// (PCIe is full duplex), so a chunk costs about max(bytes in, bytes out) / link rate.  Reads that
// the caller keeps in pinned memory (psigpu_host_alloc) are copied in place; pageable reads are
// staged through pinned buffers by a helper thread that runs ahead of the compute loop.  With
// PSIGPU_SORT_UNIQUE every sub-batch is sorted on the device; sub-batches are contiguous read
// ranges, so their concatenation is the sorted chunk.
//
// The reads of a chunk as the caller hands them over: ASCII bases back to back (psigpu_find_seeds), or 2-bit codes in
// u64 words + an optional "not ACGT" bit per base (psigpu_find_seeds_packed; layout at k_seed_pack).  read_off counts
// BASES either way.  A sub-batch [b0, b1) of bases is then a byte range of each source array: cut at word boundaries
// for the packed arrays, so that words stay 8-byte aligned on the device.
struct ReadsIn {
  const char* ascii = nullptr;
  const uint64_t* words = nullptr;
  const uint64_t* mask = nullptr;
  bool packed() const { return words != nullptr; }
  // main array (bases / 2-bit words): first byte, byte count, and the base index the first byte's first base has
  size_t main_byte0(uint64_t b0) const { return packed() ? (size_t)(b0 >> 5) * 8 : (size_t)b0; }
  size_t main_bytes(uint64_t b0, uint64_t b1) const { return b1 == b0 ? 0 : packed() ? (size_t)(((b1 + 31) >> 5) - (b0 >> 5)) * 8 : (size_t)(b1 - b0); }
  const char* main_ptr() const { return packed() ? reinterpret_cast<const char*>(words) : ascii; }
  size_t mask_byte0(uint64_t b0) const { return (size_t)(b0 >> 6) * 8; }
  size_t mask_bytes(uint64_t b0, uint64_t b1) const { return (mask && b1 != b0) ? (size_t)(((b1 + 63) >> 6) - (b0 >> 6)) * 8 : 0; }
};

static int find_seeds_host(psigpu_ctx* ctx, const ReadsIn& in, const uint64_t* read_off, uint64_t n_reads,
                           uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, psigpu_hits* out)
{
  const char* bases = in.main_ptr();
  if (!ctx || !out || (n_reads && (!read_off))) return PSIGPU_ERR_ARG;
  const auto t_entry = std::chrono::steady_clock::now();
  out->n = 0; out->data = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if ((flags & PSIGPU_ALL) == 0) flags |= PSIGPU_ALL;
  const bool want_sort = (flags & PSIGPU_SORT_UNIQUE) != 0;
  flags &= ~PSIGPU_SORT_UNIQUE;
  const uint64_t n_bases = n_reads ? read_off[n_reads] : 0;
  // ASCII reads start at bases[0]; packed reads may be a RANGE of a larger chunk (psikt --devices: every GPU takes a
  // contiguous range of the chunk's reads and the word arrays cannot be offset by a base count): read_off[0] = the
  // range's first base in the arrays
  if (n_reads && read_off[0] != 0 && !in.packed()) { ctx->err = "read_off[0] must be 0"; return PSIGPU_ERR_ARG; }
  const uint64_t base0 = n_reads ? read_off[0] : 0;
  if (base0 > n_bases) { ctx->err = "read offsets must not decrease"; return PSIGPU_ERR_ARG; }
  const uint64_t org2 = in.packed() ? (base0 & ~31ull) : 0, orgm = base0 & ~63ull;      // first bases of the device buffers
  const size_t M0 = in.main_byte0(org2), K0 = in.mask_byte0(orgm);
  if (n_bases && !bases) return PSIGPU_ERR_ARG;
  if (n_reads == 0) {                 // nothing to copy either way; the counters of the call are still set
    uint64_t n = 0;
    return run_pipeline(ctx, nullptr, nullptr, 0, 0, k, step, rec_offset, flags, nullptr, &n);
  }

  // sub-batches: cut at read boundaries; the first ones are small and double up to SUB_BYTES of bases, so
  // that the copy-out stream -- the longest stage -- starts early (what precedes its first transfer is
  // the one part of the call nothing overlaps with)
  // (PSIGPU_SUB_BYTES: tests force many sub-batches on small inputs)
  const char* sub_env = getenv("PSIGPU_SUB_BYTES");
  // (bases per sub-batch.  ASCII reads: 16 Mi, the link is the bound and more, smaller pieces hide its latency; packed reads:
  // 32 Mi -- the link moves a quarter of the bytes, the compute loop is the critical path and every sub-batch costs ~75 us
  // of launches, event records and its synchronisation: 2.30 ms per 1 M-read chunk at 16 Mi, 2.12 at 32, tools/e2e_packed.py)
  const uint64_t SUB_BYTES = ctx->opt_sub_bytes ? ctx->opt_sub_bytes : sub_env ? std::max<uint64_t>(1, strtoull(sub_env, nullptr, 10))
                                                                               : ((in.packed() ? 32ull : 16ull) << 20);
  { int st = pipeline_init(ctx); if (st != PSIGPU_OK) return st; }
  // Reads AND offsets in pinned memory (psi::Records of the shim, psikt): the chunk's reads go to one device buffer
  // and their transfers are queued up to IN_RING sub-batches ahead of the compute loop, so that the copy engine
  // never waits for the host thread between two of them (with two slots it does whenever a sub-batch's kernels,
  // sort and synchronisations take longer than its transfer); the offsets are rebased by a kernel that reads the
  // caller's array in place.  150 calls of either path alternated in one process (tools/e2e_ab3.py; the boxes are
  // shared and a call varies between 3.6 and 5.3 ms): median 4.37 ms against 4.56, minimum 3.64 against 3.80.
  ptrdiff_t pin_delta = 0, off_delta = 0, mask_delta = 0;
  const bool pinned_in = n_bases == 0 || (host_ptr_is_pinned(bases, &pin_delta) && (!in.mask || host_ptr_is_pinned(in.mask, &mask_delta)));
  const bool env_no_ahead = getenv("PSIGPU_NO_AHEAD") != nullptr;      // A/B: the two-slot path (read per call: tools/e2e_ab3.py alternates)
  const bool ahead_ok = pinned_in && n_bases && ctx->ec.ok && !env_no_ahead && !ctx->opt_no_ahead && host_ptr_is_pinned(read_off, &off_delta);
  std::vector<uint64_t> cut{ 0 };
  {
    // piece sizes in bytes of bases: SUB/8, SUB/4, SUB/2, SUB ... SUB -- small at the start: what precedes the first
    // transfer out is the one part of the call nothing overlaps with.  (Small pieces at the END as well -- so that the
    // last kernels, the last records' way out and their widening are short -- were measured and do not pay: a
    // sub-batch costs ~0.1 ms of launches and synchronisation whatever its size; tools/e2e_ab.py, three alternations
    // on one box: 4.58 ms grow-only, 4.78-4.86 with tapered ends.)
    const uint64_t small = std::max<uint64_t>(1, SUB_BYTES / 8);
    std::vector<uint64_t> pieces;
    uint64_t left = n_bases - base0;
    for (uint64_t p = small; p < SUB_BYTES && left > p; p *= 2) { pieces.push_back(p); left -= p; }
    const size_t n_head = pieces.size();
    while (left > 0) { const uint64_t p = std::min(left, SUB_BYTES); pieces.push_back(p); left -= p; }
    if (pieces.size() > n_head + 1 && pieces.back() < SUB_BYTES / 4) {      // no tiny tail
      const uint64_t t = pieces.back(); pieces.pop_back(); pieces.back() += t;
    }
    // (With the reads' transfers queued ahead -- below -- small pieces at the end were tried again: SUB/2, SUB/4, SUB/4
    // instead of the last full piece.  150 calls of each variant alternated in one process, tools/e2e_ab3.py: median
    // 4.53 ms against 4.37 without, minimum 3.56 against 3.64: not kept.)
    uint64_t target = base0;
    for (size_t pi = 0; pi < pieces.size() && cut.back() < n_reads; ++pi) {
      target += pieces[pi];
      uint64_t r = (target >= n_bases || pi + 1 == pieces.size()) ? n_reads
                                                                    : (uint64_t)(std::lower_bound(read_off, read_off + n_reads, target) - read_off);
      if (r <= cut.back()) r = cut.back() + 1;            // at least one read per sub-batch
      cut.push_back(std::min(r, n_reads));
    }
    if (cut.back() < n_reads) cut.push_back(n_reads);
  }
  const size_t n_sub = cut.size() - 1;
  const bool ahead = ahead_ok && n_sub > 1;
  constexpr size_t IN_RING = psigpu_ctx::EngineCopy::IN_RING;
  size_t issued = 0;
  auto issue_in = [&](size_t j) -> bool {
    const uint64_t b0 = read_off[cut[j]], b1 = read_off[cut[j + 1]], nb = b1 - b0;
    hsa_signal_t sg = ctx->ec.sig_in[j % IN_RING];
    if (nb == 0) { hsa_signal_store_relaxed(sg, 0); return true; }
    // (packed reads: neighbouring sub-batches share their boundary word -- both transfers write the same bytes there)
    const size_t m0 = in.main_byte0(b0), mb = in.main_bytes(b0, b1), k0 = in.mask_byte0(b0), kb = in.mask_bytes(b0, b1);
    const int n_copies = kb ? 2 : 1;
    if (!engine_copy(ctx, true, (char*)ctx->in_bases.p + (m0 - M0), bases + pin_delta + m0, mb, sg, n_copies)) return false;
    if (kb && !engine_copy_more(ctx, true, (char*)ctx->in_mask.p + (k0 - K0), (const char*)in.mask + mask_delta + k0, kb, sg)) {
      engine_wait_value(sg, 2);                  // the first transfer is on its way: wait for it, then fail
      hsa_signal_store_relaxed(sg, 0);
      return false;
    }
    return true;
  };
  if (ahead) {
    HIPCHK(ctx, ctx->in_bases.ensure(in.main_bytes(org2, n_bases) + 64));
    if (in.mask) HIPCHK(ctx, ctx->in_mask.ensure(in.mask_bytes(orgm, n_bases) + 64));
    for (; issued < std::min(n_sub, IN_RING); ++issued)
      if (!issue_in(issued)) {
        for (size_t j = 0; j < issued; ++j) engine_wait(ctx->ec.sig_in[j]);
        ctx->err = "staging the reads: engine copy failed";
        return PSIGPU_ERR_DEVICE;
      }
  }

  std::function<void(int)> trace_in;
  // staging of sub-batch j into slot j % 2 and its H2D on the copy-in stream
  auto stage_in = [&](size_t j) -> hipError_t {
    psigpu_ctx::Slot& sl = ctx->slot[j & 1];
    const uint64_t r0 = cut[j], r1 = cut[j + 1], nr = r1 - r0;
    const uint64_t b0 = read_off[r0], b1 = read_off[r1], nb = b1 - b0;
    const size_t mbytes = in.main_bytes(b0, b1), kbytes = in.mask_bytes(b0, b1);
    const size_t mstage = (mbytes + 255) & ~(size_t)255;
    const size_t off_bytes = ((nr + 1) * 8 + 255) & ~(size_t)255;
    const size_t need = off_bytes + (pinned_in ? 0 : mstage + kbytes);
    if (need > sl.h_cap) {
      if (sl.h_stage) (void)hipHostFree(sl.h_stage);
      sl.h_stage = nullptr; sl.h_cap = 0;
      hipError_t e = hipHostMalloc(&sl.h_stage, need + need / 4 + 4096, hipHostMallocMapped);
      if (e == hipSuccess) e = hipHostGetDevicePointer(&sl.h_stage_dev, sl.h_stage, 0);
      if (e != hipSuccess) return e;
      sl.h_cap = need + need / 4 + 4096;
    }
    hipError_t e = sl.bases.ensure(mbytes + 64);
    if (e == hipSuccess) e = sl.off.ensure((nr + 1) * 8);
    if (e == hipSuccess && in.mask) e = sl.mask.ensure(kbytes + 64);
    if (e != hipSuccess) return e;
    uint64_t* ho = reinterpret_cast<uint64_t*>(sl.h_stage);
    for (uint64_t r = 0; r <= nr; ++r) ho[r] = read_off[r0 + r] - b0;
    const char* src = bases + in.main_byte0(b0);
    const char* msrc = in.mask ? (const char*)in.mask + in.mask_byte0(b0) : nullptr;
    if (!pinned_in && nb) {
      parallel_copy((char*)sl.h_stage + off_bytes, src, mbytes); src = (char*)sl.h_stage + off_bytes;
      if (kbytes) { memcpy((char*)sl.h_stage + off_bytes + mstage, msrc, kbytes); msrc = (char*)sl.h_stage + off_bytes + mstage; }
    }
    // one transfer per sub-batch and array (the read offsets reach the device through k_publish from this
    // mapped staging buffer)
    if (trace_in) trace_in(0);
    if (ctx->ec.ok) {
      hsa_signal_t sg = ctx->ec.sig_in[j & 1];
      if (nb == 0) hsa_signal_store_relaxed(sg, 0);
      else {
        if (!engine_copy(ctx, true, sl.bases.p, pinned_in ? src + pin_delta : src, mbytes, sg, kbytes ? 2 : 1)) return hipErrorUnknown;
        if (kbytes && !engine_copy_more(ctx, true, sl.mask.p, pinned_in ? msrc + mask_delta : msrc, kbytes, sg)) {
          engine_wait_value(sg, 2);
          hsa_signal_store_relaxed(sg, 0);
          return hipErrorUnknown;
        }
      }
    } else {
      if (nb) e = hipMemcpyAsync(sl.bases.p, src, mbytes, hipMemcpyHostToDevice, ctx->s_in);
      if (e == hipSuccess && kbytes) e = hipMemcpyAsync(sl.mask.p, msrc, kbytes, hipMemcpyHostToDevice, ctx->s_in);
      if (e == hipSuccess) e = hipEventRecord(sl.in_ready, ctx->s_in);
    }
    if (trace_in) trace_in(1);
    return e;
  };

  // Pageable input: a helper thread stages ahead of the compute loop (its memcpy would otherwise sit
  // between the kernels and the D2H of every sub-batch).  staged = sub-batches enqueued so far,
  // consumed = sub-batches whose kernels have finished (their slot may be overwritten).
  std::atomic<size_t> staged{ 0 }, consumed{ 0 };
  std::atomic<int> stage_err{ (int)hipSuccess };
  std::atomic<bool> stop{ false };
  const bool use_thread = !pinned_in && n_sub > 1;
  if (use_thread && !ctx->stager) ctx->stager = new Worker;
  Worker* stager = static_cast<Worker*>(ctx->stager);
  if (use_thread) {
    stager->run([&] {
      if (hipSetDevice(ctx->device) != hipSuccess) { stage_err = (int)hipErrorInvalidDevice; return; }
      for (size_t j = 0; j < n_sub && !stop; ++j) {
        while (j >= consumed.load(std::memory_order_acquire) + 2 && !stop) std::this_thread::yield();
        if (stop) break;
        hipError_t e = stage_in(j);
        if (e != hipSuccess) { stage_err = (int)e; return; }
        staged.store(j + 1, std::memory_order_release);
      }
    });
  }
  struct Joiner {                                   // (every way out: the helper has left the call's buffers)
    Worker* w; bool used; std::atomic<bool>& stop;
    ~Joiner() { stop = true; if (used && w) w->wait(); }
  } joiner{ stager, use_thread, stop };

  // output: pinned, sized from what earlier calls produced per read; regrown when a chunk has more
  uint64_t out_cap = 0, done = 0;
  psigpu_hit* hp = nullptr;
  Widener* wd = nullptr;                            // set once the widener runs: hp must not move or go while it writes
  auto fail = [&](int st) {
    if (ahead) for (size_t j = 0; j < IN_RING; ++j) engine_wait(ctx->ec.sig_in[j]);      // transfers of reads still queued
    if (wd) wd->wait_finished(wd->posted.load());
    if (ctx->ec.ok) { engine_wait(ctx->ec.sig_out[0]); engine_wait(ctx->ec.sig_out[1]); for (auto& sg : ctx->ec.sig_fast) engine_wait(sg); }      // transfers into hp still in flight
    else if (ctx->s_out) (void)hipStreamSynchronize(ctx->s_out);
    if (hp) g_pinned.put(hp);
    hp = nullptr;
    return st;
  };
  auto out_reserve = [&](uint64_t want_records) -> int {
    if (want_records <= out_cap) return PSIGPU_OK;
    if (wd) wd->wait_finished(wd->posted.load());
    if (ctx->ec.ok) { engine_wait(ctx->ec.sig_out[0]); engine_wait(ctx->ec.sig_out[1]); for (auto& sg : ctx->ec.sig_fast) engine_wait(sg); }
    else HIPCHK(ctx, hipStreamSynchronize(ctx->s_out));
    const uint64_t cap2 = want_records + want_records / 4 + 4096;
    psigpu_hit* np = (psigpu_hit*)g_pinned.get(cap2 * sizeof(psigpu_hit));
    if (!np) { ctx->err = "cannot allocate pinned host memory for the hits"; return PSIGPU_ERR_NOMEM; }
    if (hp) { memcpy(np, hp, done * sizeof(psigpu_hit)); g_pinned.put(hp); }
    hp = np; out_cap = cap2;
    return PSIGPU_OK;
  };
  {
    const double ratio = ctx->hits_per_read_hint > 0 ? ctx->hits_per_read_hint * 1.1 : 10.0;
    int st = out_reserve((uint64_t)(ratio * (double)n_reads) + 1024);
    if (st != PSIGPU_OK) return fail(st);      // (reads may already be on their way in: fail() waits for them)
  }

  // 16-byte records over the link, widened on the host (k_hits_wire16): when the node ids are rank + constant
  static const bool env_no_wire = getenv("PSIGPU_NO_WIRE16") != nullptr;      // A/B: 32-byte records over the link
  static const bool env_no_wire8 = getenv("PSIGPU_NO_WIRE8") != nullptr;      // A/B: 16-byte records at the least
  const bool wire16 = ctx->id_affine && !env_no_wire && ctx->opt_wire != 32;
  // 8-byte records (k_hits_wire8) when the node fields leave the read fields enough bits; per sub-batch: its read count
  // sets the read-id bits, the read offset gets the rest
  const uint32_t w_noff_bits = bits_for(ctx->max_node_len), w_node_bits = bits_for(ctx->n_nodes ? ctx->n_nodes - 1 : 0);
  auto wire_fmt = [&](uint64_t nr_) {
    WireFmt f;
    if (!wire16) return f;
    f.bytes = 16;
    if (env_no_wire8 || ctx->opt_wire == 16 || ctx->wire8_overflowed) return f;
    const uint32_t used = w_noff_bits + w_node_bits + bits_for(nr_ ? nr_ - 1 : 0);
    if (used + 8 > 64) return f;                  // fewer than 8 bits for the read offset: not worth trying
    f.bytes = 8; f.noff_bits = w_noff_bits; f.node_bits = w_node_bits; f.roff_bits = std::min<uint32_t>(32, 64 - used);
    if (ctx->opt_wire8_roff_cap) f.roff_bits = std::min(f.roff_bits, ctx->opt_wire8_roff_cap);      // (tests: make a long read overflow)
    return f;
  };
  if (!ctx->widener) ctx->widener = new Widener;
  Widener& widener = *static_cast<Widener*>(ctx->widener);
  struct Session { Widener& w; ~Session() { w.end(); } } session{ widener };      // (every way out: its threads have left the call's buffers)
  if (wire16) {
    wd = &widener;
    const unsigned hw = std::thread::hardware_concurrency();
    widener.begin(std::max(1u, std::min(8u, hw / 4)), n_sub, [ctx](int slot_) {
      if (ctx->ec.ok) engine_wait(slot_ < 2 ? ctx->ec.sig_out[slot_] : ctx->ec.sig_fast[slot_ - 2]);      // (2 + q: the lookahead path's slot q)
      else { (void)hipSetDevice(ctx->device); (void)hipEventSynchronize(ctx->slot[slot_].out_done); }
    });
  }
  psigpu_counters acc{};
  const bool trace = getenv("PSIGPU_TRACE") != nullptr;        // per-sub-batch host timeline on stderr
  ctx->trace_call = trace;
  auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_call = now_ms();
  std::vector<double> tr;
  // device-side timeline (trace mode): events around every H2D, kernel phase and D2H
  std::vector<hipEvent_t> tev;
  hipEvent_t tev0 = nullptr;
  auto tmark = [&](hipStream_t st_) { if (!trace || ctx->ec.ok) return; hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st_); tev.push_back(e); };
  std::vector<hipEvent_t> tin;
  if (trace) {
    (void)hipEventCreate(&tev0); (void)hipEventRecord(tev0, ctx->s_comp);
    if (!ctx->ec.ok) trace_in = [&](int) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, ctx->s_in); tin.push_back(e); };
  }
  bool host_sort = false;            // some sub-batch did not fit the device sorter's key
  uint32_t wire_widest = 0;          // bytes per wire record, the widest any sub-batch used
  hipStream_t sc = ctx->s_comp;
  if (!use_thread && !ahead) {
    hipError_t e = stage_in(0);
    if (e != hipSuccess) { ctx->err = std::string("staging the reads: ") + hipGetErrorString(e); return fail(PSIGPU_ERR_DEVICE); }
  }
  // ---- two sub-batches in flight (enqueue_default): chunks whose sub-batches need the default mode's kernels alone ----
  size_t first = 0;                  // sub-batches [0, first) are done when the synchronous loop below starts
  static const bool env_no_lookahead = getenv("PSIGPU_NO_LOOKAHEAD") != nullptr;
  bool fast = ahead && wire16 && !trace && !env_no_lookahead && !ctx->opt_no_lookahead && ctx->fast_k == k &&
              ctx->fast_flags == (flags & PSIGPU_ALL) && ctx->lkt_ready && ctx->kt_ready && ctx->lkt_k == k && ctx->hits_per_read_hint > 0 &&
              !(want_sort && getenv("PSIGPU_NO_GROUPED_SORT") != nullptr);
  if (fast) {
    const uint32_t stp = step ? step : k;
    uint64_t nr_max = 0, nb_max = 0;
    for (size_t j = 0; j < n_sub; ++j) {
      nr_max = std::max(nr_max, cut[j + 1] - cut[j]);
      nb_max = std::max(nb_max, read_off[cut[j + 1]] - read_off[cut[j]]);
    }
    const uint64_t seeds_max = nb_max / stp + nr_max;
    const double ratio = ctx->hits_per_read_hint * 1.25;
    const uint64_t cap_max = (uint64_t)(ratio * (double)nr_max) + 4096;
    // everything the kernels of a sub-batch touch is sized here, once: nothing may be regrown (freed) under a sub-batch in flight
    hipError_t e = hipSuccess;
    auto need = [&](DevBuf& b, size_t bytes) { if (e == hipSuccess) e = b.ensure(bytes); };
    need(ctx->w_ctr, sizeof(DevCounters)); need(ctx->w_total, 64);
    need(ctx->w_tiles, (nr_max / SCAN_TILE + 2) * 8); need(ctx->w_seed_off, (nr_max + 1) * 8);
    if (fused_step_wanted(ctx)) need(ctx->w_tilestate, (seeds_max / KS_TILE + 66) * 8);      // (one kernel per sub-batch: keys and results stay in registers)
    else {
      need(ctx->w_seed_key, (seeds_max + 1) * 8); need(ctx->w_seed_info, (seeds_max + 1) * 8);
      need(ctx->w_seedres, (seeds_max + 16) * 16);
      need(ctx->w_iv_tiles, (WAVES_MAX + 8) * 8); need(ctx->w_iv_tiles_off, (WAVES_MAX + 8) * 8);
    }
    constexpr int NF = psigpu_ctx::N_FAST;
    for (int q = 0; q < NF; ++q) {
      psigpu_ctx::FastSlot& fs = ctx->fast[q];
      need(fs.hits, (cap_max + 1) * sizeof(psigpu_hit));
      need(fs.wire, (cap_max + 1) * 16);
      need(fs.off, (nr_max + 1) * 8);
      if (e == hipSuccess && !fs.h) {
        e = hipHostMalloc(&fs.h, sizeof(DevCounters) + 64, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer(&fs.h_dev, fs.h, 0);
        if (e == hipSuccess) e = hipEventCreate(&fs.begin);
        if (e == hipSuccess) e = hipEventCreate(&fs.done);
      }
      if (e == hipSuccess && cap_max * 16 > fs.h_wire_cap) {
        if (fs.h_wire) (void)hipHostFree(fs.h_wire);
        fs.h_wire = nullptr; fs.h_wire_cap = 0;
        e = hipHostMalloc(&fs.h_wire, cap_max * 16 + 4096, hipHostMallocDefault);
        if (e == hipSuccess) fs.h_wire_cap = cap_max * 16 + 4096;
      }
    }
    if (e != hipSuccess || seeds_max >= 0xFFFFFFF0ull) { (void)hipGetLastError(); fast = false; }      // (the synchronous loop reports what is wrong, if anything is)
    struct Pending { unsigned long long serial; bool uniform; uint64_t cap; WireFmt wf; } pend[NF];
    // Sub-batch j goes through slot j % 3: two sub-batches are in the queue while the records of a third are on their way out
    // (its transfer reads the slot's wire buffer, the widening threads its landing buffer) -- with two slots the kernels of
    // sub-batch i + 2 had to wait for the transfer of sub-batch i and the arrangement was slower than none (2.5 ms against 1.95).
    auto enqueue = [&](size_t j) -> int {
      const int q = (int)(j % NF);
      psigpu_ctx::FastSlot& fs = ctx->fast[q];
      const uint64_t r0 = cut[j], nr = cut[j + 1] - r0, b0 = read_off[r0], nb = read_off[cut[j + 1]] - b0;
      engine_wait(ctx->ec.sig_in[j % IN_RING]);                 // its reads have landed
      if (j >= (size_t)NF) { engine_wait(ctx->ec.sig_fast[q]); widener.wait_finished(j - NF + 1); }      // the slot's wire and landing buffers are free
      k_rebase_offsets<<<64, 256, 0, sc>>>(reinterpret_cast<const uint64_t*>((const char*)(read_off + r0) + off_delta), fs.off.as<uint64_t>(), nr + 1);
      PackedIn pk{ nullptr, 0, 0 };
      if (in.packed()) { pk.mask = in.mask ? ctx->in_mask.as<uint64_t>() : nullptr; pk.bias2 = b0 - org2; pk.biasm = b0 - orgm; }
      FastArgs a;
      a.d_in = in.packed() ? ctx->in_bases.as<char>() : (const char*)ctx->in_bases.p + b0;
      a.pk = in.packed() ? &pk : nullptr;
      a.d_off = fs.off.as<uint64_t>();
      a.nr = nr; a.nb = nb; a.k = k; a.step = stp; a.rec_base = rec_offset + r0;
      a.want_sort = want_sort; a.claim_uniform = (flags & PSIGPU_UNIFORM_READS) != 0;
      a.wf = wire_fmt(nr); a.wire = &fs.wire;
      a.cap = (uint64_t)(ratio * (double)nr) + 4096;
      a.slot = q;
      Pending& pd = pend[q];
      pd.cap = a.cap; pd.wf = a.wf;
      return enqueue_default(ctx, a, sc, &pd.serial, &pd.uniform);
    };
    for (size_t j = 0; fast && j < std::min<size_t>(2, n_sub); ++j) { int st = enqueue(j); if (st != PSIGPU_OK) return fail(st); }
    for (size_t i = 0; fast && i < n_sub; ++i) {
      for (; issued < std::min(n_sub, i + IN_RING); ++issued)
        if (!issue_in(issued)) { ctx->err = "staging the reads: engine copy failed"; return fail(PSIGPU_ERR_DEVICE); }
      const int q = (int)(i % NF);
      psigpu_ctx::FastSlot& fs = ctx->fast[q];
      {
        // (not hipEventSynchronize: with more work queued behind the event the runtime waits for a marker it appends to the
        // stream -- i.e. for the next sub-batch as well -- which undoes the arrangement)
        hipError_t qe;
        uint32_t spins = 0;
        while ((qe = hipEventQuery(fs.done)) == hipErrorNotReady) { if (++spins > 2000) std::this_thread::yield(); }
        if (qe != hipSuccess) { ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(qe); return fail(PSIGPU_ERR_DEVICE); }
      }
      const DevCounters& h = *reinterpret_cast<const DevCounters*>(fs.h);
      const unsigned long long wflag = *reinterpret_cast<const unsigned long long*>((const char*)fs.h + sizeof(DevCounters) + 32);
      const Pending& pd = pend[q];
      const uint64_t r0 = cut[i], nr = cut[i + 1] - r0;
      const uint64_t n = h.n_hits_tab.v;
      if (h.serial.v != pd.serial) ++ctx->stale_handbacks;
      if (h.serial.v != pd.serial || (pd.uniform && h.not_uniform.v) || n > pd.cap || (want_sort && h.not_grouped.v) || wflag) {
        // something the five kernels alone do not settle: this sub-batch and the rest of the chunk the synchronous way
        if (wflag) ctx->wire8_overflowed = true;
        if (n > pd.cap) ctx->hits_per_read_hint = std::max(ctx->hits_per_read_hint, (double)n / (double)std::max<uint64_t>(1, nr));
        ++ctx->lookahead_fallbacks;
        if (hipStreamSynchronize(sc) != hipSuccess) { ctx->err = "hipStreamSynchronize"; return fail(PSIGPU_ERR_DEVICE); }
        break;
      }
      acc.n_reads += nr; acc.n_seeds += h.n_seeds_true.v; acc.n_seeds_valid += h.n_seeds_valid.total();
      acc.n_seeds_on_path += h.n_live.total(); acc.n_hits_on_path += h.hits_on(); acc.n_hits_off_path += n - h.hits_on();
      acc.n_loci = ctx->n_loci; acc.n_locus_kmers = ctx->fast_off ? ctx->lkt_n_ent : 0; acc.n_path_kmers = ctx->kt_n_path_kmers;
      acc.ms_locus_table_build = ctx->lkt_build_ms;
      { float t = 0; (void)hipEventElapsedTime(&t, fs.begin, fs.done); acc.ms_total += t; }
      acc.sorted_in_place += want_sort ? 1u : 0u;
      if (n) {
        if (done + n > out_cap) {
          const double per_read = (double)(done + n) / (double)(cut[i + 1]);
          int st = out_reserve(std::max<uint64_t>(done + n, (uint64_t)(per_read * 1.1 * (double)n_reads) + 1024));
          if (st != PSIGPU_OK) return fail(st);
        }
        if (!engine_copy(ctx, false, fs.h_wire, fs.wire.p, n * pd.wf.bytes, ctx->ec.sig_fast[q])) {
          ctx->err = "copying the hits out: engine copy failed"; return fail(PSIGPU_ERR_DEVICE);
        }
        widener.post(i, Widener::Job{ fs.h_wire, hp + done, n, ctx->id_base, rec_offset + r0, 2 + q, pd.wf });
        wire_widest = std::max(wire_widest, pd.wf.bytes);
        done += n;
      } else {
        hsa_signal_store_relaxed(ctx->ec.sig_fast[q], 0);
        widener.post(i, Widener::Job{ nullptr, nullptr, 0, 0, 0, 2 + q, WireFmt{} });
      }
      first = i + 1;
      if (i + 2 < n_sub) { int st = enqueue(i + 2); if (st != PSIGPU_OK) return fail(st); }
    }
  }
  const size_t n_lookahead = first;
  for (size_t i = first; i < n_sub; ++i) {
    psigpu_ctx::Slot& sl = ctx->slot[i & 1];
    if (use_thread) {
      while (staged.load(std::memory_order_acquire) <= i && stage_err.load() == (int)hipSuccess) std::this_thread::yield();
      if (stage_err.load() != (int)hipSuccess) {
        ctx->err = std::string("staging the reads: ") + hipGetErrorString((hipError_t)stage_err.load());
        return fail(PSIGPU_ERR_DEVICE);
      }
    } else if (ahead) {
      // (signal j % IN_RING was last used by sub-batch j - IN_RING, waited for below IN_RING iterations ago)
      for (; issued < std::min(n_sub, i + IN_RING); ++issued)
        if (!issue_in(issued)) { ctx->err = "staging the reads: engine copy failed"; return fail(PSIGPU_ERR_DEVICE); }
    } else if (i + 1 < n_sub) {
      hipError_t e = stage_in(i + 1);           // slot (i + 1) & 1: its last user, sub-batch i - 1, has been synchronised
      if (e != hipSuccess) { ctx->err = std::string("staging the reads: ") + hipGetErrorString(e); return fail(PSIGPU_ERR_DEVICE); }
    }
    const uint64_t r0 = cut[i], nr = cut[i + 1] - r0, nb = read_off[cut[i + 1]] - read_off[r0];
    if (ahead) {
      if (sl.off.ensure((nr + 1) * 8) != hipSuccess) { ctx->err = "out of device memory (read offsets)"; return fail(PSIGPU_ERR_NOMEM); }
      engine_wait(ctx->ec.sig_in[i % IN_RING]);
      if (i >= 2) engine_wait(ctx->ec.sig_out[i & 1]);
    } else if (ctx->ec.ok) {
      engine_wait(ctx->ec.sig_in[i & 1]);
      if (i >= 2) engine_wait(ctx->ec.sig_out[i & 1]);    // the hit buffer about to be written was the source of the D2H two sub-batches ago
    } else {
      if (hipStreamWaitEvent(sc, sl.in_ready, 0) != hipSuccess) { ctx->err = "hipStreamWaitEvent"; return fail(PSIGPU_ERR_DEVICE); }
      if (i >= 2 && hipStreamWaitEvent(sc, sl.out_done, 0) != hipSuccess) { ctx->err = "hipStreamWaitEvent"; return fail(PSIGPU_ERR_DEVICE); }
    }
    uint64_t n = 0;
    if (trace) { tr.push_back(now_ms() - t_call); tmark(sc); }
    if (ahead)
      k_rebase_offsets<<<64, 256, 0, sc>>>(reinterpret_cast<const uint64_t*>((const char*)(read_off + r0) + off_delta), sl.off.as<uint64_t>(), nr + 1);
    else
      k_publish<<<64, 256, 0, sc>>>(reinterpret_cast<const uint4*>(sl.h_stage_dev), sl.off.as<uint4>(), (uint32_t)(((nr + 1) * 8 + 15) / 16));
    if (wire16 && i >= 2) widener.wait_finished(i - 1);          // the slot's landing buffer: job i - 2 has been widened
    // where the sub-batch's bases lie on the device: behind the chunk's earlier ones (transfers queued ahead: one
    // buffer for the chunk) or at the start of the slot's buffer -- for packed reads from the word that holds b0
    const uint64_t b0_ = read_off[r0];
    PackedIn pk{ nullptr, 0, 0 };
    if (in.packed()) {
      pk.mask = in.mask ? (ahead ? ctx->in_mask.as<uint64_t>() : sl.mask.as<uint64_t>()) : nullptr;
      pk.bias2 = ahead ? b0_ - org2 : (b0_ & 31);
      pk.biasm = ahead ? b0_ - orgm : (b0_ & 63);
    }
    const char* d_in = in.packed() ? (ahead ? ctx->in_bases.as<char>() : sl.bases.as<char>())
                                   : (ahead ? (const char*)ctx->in_bases.p + b0_ : sl.bases.as<char>());
    int st = run_pipeline(ctx, d_in, sl.off.as<uint64_t>(), nr, nb, k, step, rec_offset + r0,
                          flags | ((want_sort && !host_sort && getenv("PSIGPU_NO_GROUPED_SORT") == nullptr) ? PSIGPU_SORT_UNIQUE : 0u), sc, &n,
                          wire16 ? &sl.d_wire : nullptr, in.packed() ? &pk : nullptr, wire_fmt(nr));
    WireFmt wf = wire_fmt(nr);
    if (wf.bytes == 8 && ctx->wire_used == 16) wf = WireFmt{ 16, 0, 0, 0 };      // (the call fell back: a field did not fit)
    if (st != PSIGPU_OK) return fail(st);
    if (trace) tr.push_back(now_ms() - t_call);
    consumed.store(i + 1, std::memory_order_release);
    const psigpu_counters pc = ctx->last;
    const psigpu_hit* src = ctx->w_hits.as<psigpu_hit>();
    const int grouped_before = ctx->grouped_state;      // 1: run_pipeline ordered each seed's hits and that was all
    float ms_sort = 0.f;
    if (want_sort && !host_sort && n) {
      uint64_t nu = 0;
      bool in_place = false;
      st = device_sort_unique(ctx, n, nr, rec_offset + r0, ctx->w_sorted[i & 1], sc, &nu, &in_place);
      if (st == PSIGPU_ERR_FORMAT) host_sort = true;
      else if (st != PSIGPU_OK) return fail(st);
      else { n = nu; if (!in_place) src = ctx->w_sorted[i & 1].as<psigpu_hit>(); ms_sort = ctx->last.ms_sort; }
    }
    bool wire_now = wire16 && !host_sort;
    // the wire records run_pipeline left are those of w_hits as it was at the end of the call: still good unless
    // the records were sorted afterwards (the radix route writes elsewhere; device_sort_unique's own in-place pass
    // reorders w_hits)
    const bool sorted_later = want_sort && !host_sort && n && grouped_before != 1;
    if (wire_now && n && (src != ctx->w_hits.as<psigpu_hit>() || sorted_later)) {
      // their wire form is made now (one more synchronisation)
      hipError_t e = sl.d_wire.ensure((n + 1) * 16);
      if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return fail(PSIGPU_ERR_DEVICE); }
      unsigned long long* h_wflag = reinterpret_cast<unsigned long long*>((char*)ctx->h_pinned + sizeof(DevCounters) + 32);
      if (wf.bytes == 8) {
        *h_wflag = 0;
        k_hits_wire8<<<2048, 256, 0, sc>>>(src, nullptr, nullptr, n, n, ctx->id_base, rec_offset + r0, wf, sl.d_wire.as<uint64_t>(),
                                           reinterpret_cast<unsigned long long*>((char*)ctx->h_pinned_dev + sizeof(DevCounters) + 32));
        if (hipStreamSynchronize(sc) != hipSuccess) { ctx->err = "hipStreamSynchronize"; return fail(PSIGPU_ERR_DEVICE); }
        if (*h_wflag) { ctx->wire8_overflowed = true; wf = WireFmt{ 16, 0, 0, 0 }; }
      }
      if (wf.bytes == 16) {
        k_hits_wire16<<<2048, 256, 0, sc>>>(src, nullptr, nullptr, n, n, ctx->id_base, rec_offset + r0, sl.d_wire.as<uint4>());
        if (hipStreamSynchronize(sc) != hipSuccess) { ctx->err = "hipStreamSynchronize"; return fail(PSIGPU_ERR_DEVICE); }
      }
    }
    if (wire_now && n) wire_widest = std::max(wire_widest, wf.bytes);
    if (n) {
      // extrapolate from the reads seen so far when the reservation turns out too small
      if (done + n > out_cap) {
        const double per_read = (double)(done + n) / (double)(cut[i + 1]);
        st = out_reserve(std::max<uint64_t>(done + n, (uint64_t)(per_read * 1.1 * (double)n_reads) + 1024));
        if (st != PSIGPU_OK) return fail(st);
      }
      if (trace) { tr.push_back(now_ms() - t_call); tmark(sc); tmark(ctx->s_out); }
      hipError_t e = hipSuccess;
      void* h_dst = hp + done;
      const void* d_src = src;
      size_t bytes = n * sizeof(psigpu_hit);
      if (wire_now) {
        if (n * 16 > sl.h_wire_cap) {
          if (sl.h_wire) (void)hipHostFree(sl.h_wire);
          sl.h_wire = nullptr; sl.h_wire_cap = 0;
          const size_t want = n * 16 + n * 4 + 4096;
          e = hipHostMalloc(&sl.h_wire, want, hipHostMallocDefault);
          if (e == hipSuccess) sl.h_wire_cap = want;
        }
        h_dst = sl.h_wire; d_src = sl.d_wire.p; bytes = n * wf.bytes;
      }
      if (e != hipSuccess) { /* reported below */ }
      else if (ctx->ec.ok) {
        if (!engine_copy(ctx, false, h_dst, d_src, bytes, ctx->ec.sig_out[i & 1])) e = hipErrorUnknown;
      } else {
        e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->s_out);
        if (e == hipSuccess) e = hipEventRecord(sl.out_done, ctx->s_out);
      }
      if (wire_now && e == hipSuccess)
        widener.post(i, Widener::Job{ sl.h_wire, hp + done, n, ctx->id_base, rec_offset + r0, (int)(i & 1), wf });
      if (trace) { tr.push_back(now_ms() - t_call); tmark(ctx->s_out); }
      if (e != hipSuccess) { ctx->err = std::string("copying the hits out: ") + hipGetErrorString(e); return fail(PSIGPU_ERR_DEVICE); }
      done += n;
    } else {
      if (ctx->ec.ok) hsa_signal_store_relaxed(ctx->ec.sig_out[i & 1], 0);
      else if (hipEventRecord(sl.out_done, ctx->s_out) != hipSuccess) { ctx->err = "hipEventRecord"; return fail(PSIGPU_ERR_DEVICE); }
    }
    if (wire16 && !(wire_now && n)) widener.post(i, Widener::Job{ nullptr, nullptr, 0, 0, 0, (int)(i & 1), WireFmt{} });      // (jobs and sub-batches count alike)
    if (src == ctx->w_hits.as<psigpu_hit>()) std::swap(ctx->w_hits, ctx->w_hits_alt);    // the next sub-batch writes the other buffer
    // counters of the chunk = sums over its sub-batches
    acc.n_reads += pc.n_reads; acc.n_seeds += pc.n_seeds; acc.n_seeds_valid += pc.n_seeds_valid;
    acc.n_seeds_on_path += pc.n_seeds_on_path; acc.n_hits_on_path += pc.n_hits_on_path;
    acc.n_hits_off_path += pc.n_hits_off_path; acc.n_kpaths += pc.n_kpaths; acc.n_spilled += pc.n_spilled;
    acc.n_lf_steps += pc.n_lf_steps; acc.n_rows_verified += pc.n_rows_verified; acc.n_locate_steps += pc.n_locate_steps;
    acc.n_loci = pc.n_loci; acc.n_locus_kmers = pc.n_locus_kmers; acc.n_path_kmers = pc.n_path_kmers;
    acc.n_loci_traversed = pc.n_loci_traversed; acc.ms_locus_table_build = pc.ms_locus_table_build;
    acc.ms_pack += pc.ms_pack; acc.ms_table += pc.ms_table; acc.ms_search += pc.ms_search; acc.ms_locate += pc.ms_locate;
    acc.ms_traverse += pc.ms_traverse; acc.ms_probe += pc.ms_probe; acc.ms_total += pc.ms_total + ms_sort; acc.ms_sort += ms_sort;
    acc.search_launches += pc.search_launches; acc.traverse_launches += pc.traverse_launches;
    acc.sorted_in_place += ctx->last.sorted_in_place;
  }
  if (ctx->ec.ok) { engine_wait(ctx->ec.sig_out[0]); engine_wait(ctx->ec.sig_out[1]); for (auto& sg : ctx->ec.sig_fast) engine_wait(sg); }
  else if (hipStreamSynchronize(ctx->s_out) != hipSuccess) { ctx->err = "hipStreamSynchronize (copy-out stream)"; return fail(PSIGPU_ERR_DEVICE); }
  if (wire16) widener.wait_finished(n_sub);
  if (trace) {
    fprintf(stderr, "[psigpu] host entry: %.3f ms from entry to the first sub-batch\n",
            t_call - std::chrono::duration<double, std::milli>(t_entry.time_since_epoch()).count());
    fprintf(stderr, "[psigpu] host entry: %zu sub-batches, %.3f ms; per sub-batch (ms since call): pipeline begin, end, D2H enqueue begin, end\n", n_sub, now_ms() - t_call);
    for (size_t j = 0; j + 3 < tr.size(); j += 4) fprintf(stderr, "[psigpu]   %.3f %.3f %.3f %.3f\n", tr[j], tr[j + 1], tr[j + 2], tr[j + 3]);
    (void)hipDeviceSynchronize();
    fprintf(stderr, "[psigpu] device timeline (ms since call): kernels begin, kernels end, D2H begin, D2H end\n");
    for (size_t j = 0; j + 3 < tev.size(); j += 4) {
      float a = 0, b = 0, c = 0, d = 0;
      (void)hipEventElapsedTime(&a, tev0, tev[j]); (void)hipEventElapsedTime(&b, tev0, tev[j + 1]);
      (void)hipEventElapsedTime(&c, tev0, tev[j + 2]); (void)hipEventElapsedTime(&d, tev0, tev[j + 3]);
      fprintf(stderr, "[psigpu]   %.3f %.3f %.3f %.3f\n", a, b, c, d);
    }
    fprintf(stderr, "[psigpu] H2D (ms since call): begin, end\n");
    for (size_t j = 0; j + 1 < tin.size(); j += 2) {
      float a = 0, b = 0;
      (void)hipEventElapsedTime(&a, tev0, tin[j]); (void)hipEventElapsedTime(&b, tev0, tin[j + 1]);
      fprintf(stderr, "[psigpu]   %.3f %.3f\n", a, b);
    }
    for (auto e : tev) (void)hipEventDestroy(e);
    for (auto e : tin) (void)hipEventDestroy(e);
    (void)hipEventDestroy(tev0);
  }
  if (want_sort && host_sort && done) done = sort_unique_hits(hp, done);      // records wider than the device key
  if (n_reads) ctx->hits_per_read_hint = std::max(ctx->hits_per_read_hint * 0.9, (double)done / (double)n_reads);
  acc.n_hits = done;
  acc.wire_bytes_per_hit = (wire16 && !host_sort && wire_widest) ? wire_widest : 32u;
  acc.lookahead_subbatches = (uint32_t)n_lookahead;
  ctx->last = acc;
  if (done == 0) { if (hp) g_pinned.put(hp); hp = nullptr; }
  out->data = hp;
  out->n = done;
  return PSIGPU_OK;
}

int psigpu_find_seeds(psigpu_ctx* ctx, const char* bases, const uint64_t* read_off, uint64_t n_reads,
                      uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, psigpu_hits* out)
{
  if (ctx && ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  ReadsIn in;
  in.ascii = bases;
  return find_seeds_host(ctx, in, read_off, n_reads, k, step, rec_offset, flags, out);
}

int psigpu_find_seeds_packed(psigpu_ctx* ctx, const uint64_t* packed, const uint64_t* n_mask, const uint64_t* read_off,
                             uint64_t n_reads, uint32_t k, uint32_t step, uint64_t rec_offset, uint32_t flags, psigpu_hits* out)
{
  if (n_reads && read_off && read_off[n_reads] && !packed) return PSIGPU_ERR_ARG;
  if (ctx && ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  ReadsIn in;
  static const uint64_t none = 0;
  in.words = packed ? packed : &none;           // (a chunk without bases: still "packed")
  in.mask = n_mask;
  return find_seeds_host(ctx, in, read_off, n_reads, k, step, rec_offset, flags, out);
}

void psigpu_free_hits(psigpu_hits* hits)
{
  if (!hits) return;
  if (hits->data) g_pinned.put(hits->data);
  hits->data = nullptr;
  hits->n = 0;
}

int psigpu_verify_resident(psigpu_ctx* ctx, uint32_t* n_changed, char* report, uint64_t report_cap)
{
  if (!ctx || !n_changed) return PSIGPU_ERR_ARG;
  if (ctx->dp_count) { ctx->err = "chunks were begun and not ended (psigpu_find_seeds_device_end)"; return PSIGPU_ERR_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());
  *n_changed = 0;
  std::string rep;
  for (const auto& r : ctx->resident) {
    uint64_t sum = 0;
    // (an array released or regrown since it was noted is "changed" without a kernel reading what may be freed memory)
    const bool moved = r.buf->p != r.at || r.bytes > r.buf->cap;
    if (!moved) { int st = device_checksum(ctx, r.at, r.bytes, &sum); if (st != PSIGPU_OK) return st; }
    if (moved || sum != r.sum) {
      ++*n_changed;
      if (!rep.empty()) rep += "; ";
      rep += r.name + " (" + std::to_string(r.bytes) + " bytes)";
    }
  }
  if (report && report_cap) {
    const size_t m = std::min<size_t>(rep.size(), (size_t)report_cap - 1);
    memcpy(report, rep.data(), m);
    report[m] = 0;
  }
  return PSIGPU_OK;
}

int psigpu_get_counters(const psigpu_ctx* ctx, psigpu_counters* out)
{
  if (!ctx || !out) return PSIGPU_ERR_ARG;
  *out = ctx->last;
  out->stale_handbacks = ctx->stale_handbacks;
  out->lookahead_fallbacks = ctx->lookahead_fallbacks;
  return PSIGPU_OK;
}

}  // extern "C"
