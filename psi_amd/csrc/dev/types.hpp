// Records, views, counters and the small device helpers every kernel uses -- part of the one translation unit device.hip (included there, in order; not a header of its own).

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
  PaddedCounter n_pfx_surv;      // k_pfx_filter: prefix walks of the starting loci that pass the chunk's prefix maps (k_traverse's roots)
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

