// K1, level-synchronous: the FM backward search as sorted sweeps over the rank blocks -- part of the one translation unit device.hip (included there, in order; not a header of its own).
// ------------------------------------------------------------------------------------
// The reference extends a pattern one character at a time (index_iter.hpp:835-841 -> Iter::go_down, fmindex.hpp:851-869
// -> sdsl::backward_search): l' = C[c] + rank_c(l), r' = C[c] + rank_c(r).  k_fm_search (dev/k_fm_search.hpp) does that
// with one quad per seed: every step of every seed is an independent random 64-byte sector (round 5: 41 M of them per
// 1 M-read chunk, 2.6 GB fetched for a 17-MB rank structure).
//
// Here all seeds take the SAME step at the same time, and they take it in the order of their interval: for a fixed
// character LF is monotone in the row, so seeds that stand close together in the suffix array and read the same
// character stand close together afterwards.  A ROUND is
//   partition   the live seeds into 4096 buckets by where their interval begins (round 0: by the leading bases of the
//               q-mer the interval table is indexed with -- the same order), count / scan / scatter: streams;
//   sweep       one workgroup per bucket.  The bucket's seeds lie in one narrow range of rows [min l, max r]: its rank
//               blocks are STAGED IN LDS ONCE and every seed of the bucket takes its step from there (7 M seeds over
//               270 K blocks: ~26 seeds share a block).  The four images of the range under LF are narrow again -- the
//               workgroup follows the tree of ranges (4, 16, 64, 256 nodes), staging each level's blocks once, for
//               SW_LEVELS steps; a few more steps (the tail of the seed) go to memory directly; then the intervals that
//               are still alive are compacted (wave ballot + prefix) to the front of the bucket's region for the next
//               round, or, in the last round, leave as (first row, count) per seed.
// No seed ever waits for another seed's sector; a rank block travels from HBM once per round, not once per seed.
// One-word seeds (k <= 31).  Records are 16 bytes: interval, seed number, the next sixteen characters.
// ------------------------------------------------------------------------------------
#ifndef SW_BUCKET_BITS_V
#define SW_BUCKET_BITS_V 12
#endif
#ifndef SW_PER_V
#define SW_PER_V 4
#endif
#ifndef SW_CAP_V
#define SW_CAP_V 320
#endif
constexpr uint32_t SW_BUCKET_BITS = SW_BUCKET_BITS_V, SW_BUCKETS = 1u << SW_BUCKET_BITS;      // (measured: tools/r06_sweep_ab.sh)
constexpr uint32_t SW_PER = SW_PER_V, SW_TILE = 256 * SW_PER;      // seeds of a bucket a workgroup holds in registers at a time
constexpr uint32_t SW_LEVELS = 5;                           // steps per round answered from LDS (tree of 1 + 4 + 16 + 64 + 256 ranges)
constexpr uint32_t SW_NODES = 256;                          // 4^(SW_LEVELS - 1)
constexpr uint32_t SW_CAP = SW_CAP_V;                       // rank blocks staged at a time (64 bytes each)
constexpr uint32_t SW_MINI = 6;                             // no interval table in the index: one of 4^6 entries is made when the part is first searched
constexpr uint32_t SW_DEAD = 0xFFFFFFFFu;                   // record field l: no interval (a hole in a bucket's region)
constexpr uint32_t SW_KR_LEVELS = 16;
// four workgroups per CU (128 VGPRs, 36 KB of LDS): a level is a chain of barriers and one memory round trip, and what hides it is
// another workgroup -- 0.84 -> 0.78 ms after the table, 1.48 -> 1.34 without against three per CU; five (96 VGPRs) spill and lose it again
#ifndef SW_OCCUPANCY_ATTR
#define SW_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif                       // characters a record carries

// rank_c(i) by ONE lane from a rank block it can address (LDS or memory): the same arithmetic as quad_rank
template <typename P>
__device__ __forceinline__ uint32_t block_rank(const FMView& fm, const uint32_t* s_sup, P b /* 4 x 16 bytes */, uint32_t c, uint32_t i)
{
  const uint32_t blk = i / BLOCK_SYMS, off = i - blk * BLOCK_SYMS;
  const uint4 h = b[0];
  uint32_t base;
  if (c == 3) base = blk * BLOCK_SYMS - h.x - h.y - h.z - (h.w >> 8) - exc_super(fm, s_sup, blk);
  else base = c == 0 ? h.x : c == 1 ? h.y : h.z;
  if (c == 0 && (h.w & 0xFF) != 0) {
    const uint32_t e = (h.w >> 8) + exc_super(fm, s_sup, blk), ne = h.w & 0xFF;
    base -= exc_below(fm.exc_row + e, ne == 255 ? fm.n_exc - e : ne, i);
  }
  const uint32_t g = off >> 6, m = off & 63u;
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t gi = 0; gi < 3; ++gi) {
    if (gi > g || (gi == g && m == 0)) continue;
    const uint4 v = b[1 + gi];
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint64_t eq = (lo ^ ((c & 1u) ? 0ull : ~0ull)) & (hi ^ ((c & 2u) ? 0ull : ~0ull));
    cnt += (uint32_t)__popcll(gi < g ? eq : (eq & ((1ull << m) - 1ull)));
  }
  return base + cnt;
}

// both ends of an interval: one set of loads when they lie in the same block (the rule once the interval is small)
template <typename P>
__device__ __forceinline__ void block_rank_pair(const FMView& fm, const uint32_t* s_sup, P bl, P br, bool same, uint32_t c, uint32_t l, uint32_t r,
                                                uint32_t& nl, uint32_t& nr)
{
  if (!same) { nl = block_rank(fm, s_sup, bl, c, l); nr = block_rank(fm, s_sup, br, c, r); return; }
  const uint32_t blk = l / BLOCK_SYMS, ol = l - blk * BLOCK_SYMS, orr = r - blk * BLOCK_SYMS;
  const uint4 h = bl[0];
  uint32_t base;
  if (c == 3) base = blk * BLOCK_SYMS - h.x - h.y - h.z - (h.w >> 8) - exc_super(fm, s_sup, blk);
  else base = c == 0 ? h.x : c == 1 ? h.y : h.z;
  uint32_t el = 0, er = 0;
  if (c == 0 && (h.w & 0xFF) != 0) {
    const uint32_t e = (h.w >> 8) + exc_super(fm, s_sup, blk), ne = h.w & 0xFF;
    el = exc_below(fm.exc_row + e, ne == 255 ? fm.n_exc - e : ne, l);
    er = exc_below(fm.exc_row + e, ne == 255 ? fm.n_exc - e : ne, r);
  }
  uint32_t cl = 0, cr = 0;
#pragma unroll
  for (uint32_t gi = 0; gi < 3; ++gi) {
    if (orr <= gi * 64) continue;                 // (r >= l: nothing of this group lies below either)
    const uint4 v = bl[1 + gi];
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint64_t eq = (lo ^ ((c & 1u) ? 0ull : ~0ull)) & (hi ^ ((c & 2u) ? 0ull : ~0ull));
    const uint32_t ml = ol > gi * 64 ? min(ol - gi * 64, 64u) : 0u, mr = min(orr - gi * 64, 64u);
    cl += (uint32_t)__popcll(ml >= 64 ? eq : (eq & ((1ull << ml) - 1ull)));
    cr += (uint32_t)__popcll(mr >= 64 ? eq : (eq & ((1ull << mr) - 1ull)));
  }
  nl = base - el + cl; nr = base - er + cr;
}

// ... from a STAGED block: its 64 bytes in LDS plus, per half-group of 32 symbols, how many A / C / G / T lie in front of it
// inside the block (s_cum: six words per block, a byte per character; made once per block by the lanes that staged its
// groups -- k_fm_sweep's staging): a rank is then ONE 32-bit popcount, where a block read from memory costs up to three
// 64-bit ones over two bit planes.  The sweep is bound by its VALU work (281 M wave instructions per round of five steps
// with three popcounts per rank, 176 M with one 64-bit popcount, measured; 0.29 ms of pure issue), and ~26 seeds share a
// block's counts.
__device__ __forceinline__ uint32_t staged_base(const FMView& fm, const uint32_t* s_sup, const uint4 h, uint32_t c, uint32_t blk)
{
  if (c == 3) return blk * BLOCK_SYMS - h.x - h.y - h.z - (h.w >> 8) - exc_super(fm, s_sup, blk);
  return c == 0 ? h.x : c == 1 ? h.y : h.z;
}
// symbols of character c among the first `off` of the block in `slot`
__device__ __forceinline__ uint32_t staged_in_block(const uint4* s_blk, const uint32_t* s_cum, uint32_t slot, uint32_t c, uint32_t off)
{
  const uint32_t h = off >> 5, m = off & 31u;
  const uint32_t pre = (s_cum[slot * 6 + h] >> (8 * c)) & 0xFFu;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(s_blk) + slot * 16 + 4 + (h >> 1) * 4 + (h & 1u);      // the group's planes: lo lo hi hi
  const uint32_t lo = w[0], hi = w[2];
  const uint32_t eq = (lo ^ ((c & 1u) ? 0u : ~0u)) & (hi ^ ((c & 2u) ? 0u : ~0u));
  return pre + (uint32_t)__popc(eq & ((1u << m) - 1u));
}
__device__ __forceinline__ uint32_t staged_rank(const FMView& fm, const uint32_t* s_sup, const uint4* s_blk, const uint32_t* s_cum, uint32_t slot,
                                                uint32_t c, uint32_t i)
{
  const uint32_t blk = i / BLOCK_SYMS;
  const uint4 hd = s_blk[slot * 4];
  uint32_t base = staged_base(fm, s_sup, hd, c, blk);
  if (c == 0 && (hd.w & 0xFF) != 0) {
    const uint32_t e = (hd.w >> 8) + exc_super(fm, s_sup, blk), ne = hd.w & 0xFF;
    base -= exc_below(fm.exc_row + e, ne == 255 ? fm.n_exc - e : ne, i);
  }
  return base + staged_in_block(s_blk, s_cum, slot, c, i - blk * BLOCK_SYMS);
}
__device__ __forceinline__ void staged_rank_pair(const FMView& fm, const uint32_t* s_sup, const uint4* s_blk, const uint32_t* s_cum, uint32_t slot_l,
                                                 uint32_t slot_r, uint32_t c, uint32_t l, uint32_t r, uint32_t& nl, uint32_t& nr)
{
  nl = staged_rank(fm, s_sup, s_blk, s_cum, slot_l, c, l);
  if (slot_r == slot_l) {
    // both ends in one block (the rule once the interval is small): the header's part is shared, the second end adds its own symbols
    const uint32_t blk = l / BLOCK_SYMS, ol = l - blk * BLOCK_SYMS, orr = r - blk * BLOCK_SYMS;
    uint32_t d = staged_in_block(s_blk, s_cum, slot_l, c, orr) - staged_in_block(s_blk, s_cum, slot_l, c, ol);
    const uint4 hd = s_blk[slot_l * 4];
    if (c == 0 && (hd.w & 0xFF) != 0) {
      const uint32_t e = (hd.w >> 8) + exc_super(fm, s_sup, blk), ne = hd.w & 0xFF;
      const uint32_t nn = ne == 255 ? fm.n_exc - e : ne;
      d -= exc_below(fm.exc_row + e, nn, r) - exc_below(fm.exc_row + e, nn, l);
    }
    nr = nl + d;
  } else nr = staged_rank(fm, s_sup, s_blk, s_cum, slot_r, c, r);
}

struct SweepPart {          // how a round's records are partitioned
  uint32_t q0;              // round 0: bases of the q-mer the interval table is indexed with
  uint32_t shift;           // rounds >= 1: bucket = first row >> shift
  uint32_t n_wg;            // workgroups of the count / scatter kernels
};
__device__ __forceinline__ uint32_t sw_bucket_key(uint64_t key, const SweepPart& sp)
{
  const uint32_t qm = (uint32_t)(key & ((1ull << (2 * sp.q0)) - 1ull));
  return 2 * sp.q0 > SW_BUCKET_BITS ? qm >> (2 * sp.q0 - SW_BUCKET_BITS) : qm;
}
__device__ __forceinline__ uint32_t sw_bucket_row(uint32_t l, const SweepPart& sp) { return min(SW_BUCKETS - 1, l >> sp.shift); }

// count: BY_KEY from the chunk's seeds (round 0), else from the records the previous round left behind (*n_rec of them)
template <bool BY_KEY>
__global__ void __launch_bounds__(256)
k_sweep_count(const uint64_t* __restrict__ seed_key, const uint4* __restrict__ rec, const uint64_t* __restrict__ n_ptr, uint64_t cap, SweepPart sp,
              uint32_t* __restrict__ cnt /* [bucket][wg] */)
{
  __shared__ uint32_t hist[SW_BUCKETS];
  for (uint32_t i = threadIdx.x; i < SW_BUCKETS; i += 256) hist[i] = 0;
  __syncthreads();
  const uint64_t n = min(*n_ptr, cap);
  const uint64_t s0 = (uint64_t)blockIdx.x * SB_TILE, s1 = min(n, s0 + SB_TILE);
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += 256 * 8) {
    if (BY_KEY) {
      uint64_t key[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) key[j] = s + 256 * j < s1 ? seed_key[s + 256 * j] : KEY_INVALID;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (key[j] != KEY_INVALID) atomicAdd(&hist[sw_bucket_key(key[j], sp)], 1u);
    } else {
      uint32_t l[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) l[j] = s + 256 * j < s1 ? rec[s + 256 * j].x : SW_DEAD;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (l[j] != SW_DEAD) atomicAdd(&hist[sw_bucket_row(l[j], sp)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < SW_BUCKETS; i += 256) cnt[(uint64_t)i * sp.n_wg + blockIdx.x] = hist[i];
}

// scatter: a record per live seed into its bucket's region.  Round 0 makes the record: (q-mer, -, seed, next 16 characters);
// `refill_lvl` != 0: the records' characters are used up -- the next sixteen from the seed itself (seeds of more than q0 + 16 bases)
template <bool BY_KEY>
__global__ void __launch_bounds__(256)
k_sweep_scatter(const uint64_t* __restrict__ seed_key, const uint4* __restrict__ rec, const uint64_t* __restrict__ n_ptr, uint64_t cap, SweepPart sp,
                const uint64_t* __restrict__ base /* first record of every bucket */, const uint32_t* __restrict__ within /* [bucket][wg] */,
                uint32_t refill_lvl, uint4* __restrict__ out)
{
  __shared__ uint32_t cur[SW_BUCKETS];
  for (uint32_t i = threadIdx.x; i < SW_BUCKETS; i += 256) cur[i] = (uint32_t)base[i] + within[(uint64_t)i * sp.n_wg + blockIdx.x];
  __syncthreads();
  const uint64_t n = min(*n_ptr, cap);
  const uint64_t s0 = (uint64_t)blockIdx.x * SB_TILE, s1 = min(n, s0 + SB_TILE);
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += 256 * 8) {
    if (BY_KEY) {
      uint64_t key[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) key[j] = s + 256 * j < s1 ? seed_key[s + 256 * j] : KEY_INVALID;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (key[j] == KEY_INVALID) continue;
        const uint32_t at = atomicAdd(&cur[sw_bucket_key(key[j], sp)], 1u);
        out[at] = make_uint4((uint32_t)(key[j] & ((1ull << (2 * sp.q0)) - 1ull)), 0u, (uint32_t)(s + 256 * j), (uint32_t)(key[j] >> (2 * sp.q0)));
      }
    } else {
      uint4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = s + 256 * j < s1 ? rec[s + 256 * j] : make_uint4(SW_DEAD, 0, 0, 0);
      if (refill_lvl) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (r[j].x != SW_DEAD) r[j].w = (uint32_t)(seed_key[r[j].z] >> (2 * refill_lvl));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (r[j].x == SW_DEAD) continue;
        const uint32_t at = atomicAdd(&cur[sw_bucket_row(r[j].x, sp)], 1u);
        out[at] = r[j];
      }
    }
  }
}

// offsets of the (bucket, workgroup) counts: cnt[b][w] -> where workgroup w's records of bucket b start.  One workgroup per
// bucket turns its row into exclusive prefixes and leaves the row's total; one workgroup scans the totals; the consumers
// add the two (sw_off).  (The three-kernel scan over the whole matrix -- k_scan_tiles / sums / final -- took 50 us per round for
// 1.7 M counters: a quarter of the partition.)
__global__ void __launch_bounds__(256)
k_sweep_rows(uint32_t* __restrict__ cnt, uint32_t n_wg, uint32_t* __restrict__ row_total)
{
  __shared__ uint32_t s_w[4];
  uint32_t* row = cnt + (uint64_t)blockIdx.x * n_wg;
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n_wg; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < n_wg ? row[i] : 0u;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, d); if (lane >= (uint32_t)d) incl += u; }
    __syncthreads();
    if (lane == 63) s_w[wib] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) { if (w < wib) before += s_w[w]; all += s_w[w]; }
    if (i < n_wg) row[i] = carry + before + incl - v;
    carry += all;
  }
  if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}
__global__ void __launch_bounds__(1024)
k_sweep_bases(const uint32_t* __restrict__ row_total, uint32_t n_rows, uint64_t* __restrict__ base /* [n_rows + 1] */)
{
  __shared__ uint64_t s_w[16];
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  uint64_t carry = 0;
  for (uint32_t b0 = 0; b0 < n_rows; b0 += 1024) {
    const uint32_t i = b0 + threadIdx.x;
    const uint64_t v = i < n_rows ? row_total[i] : 0ull;
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t u = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += u; }
    __syncthreads();
    if (lane == 63) s_w[wib] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (uint32_t w = 0; w < 16; ++w) { if (w < wib) before += s_w[w]; all += s_w[w]; }
    if (i < n_rows) base[i] = carry + before + incl - v;
    carry += all;
  }
  if (threadIdx.x == 0) base[n_rows] = carry;
}

// the interval of every q-mer, q = SW_MINI (or the seed length when that is shorter), by q LF steps each: what round 0 looks
// a seed's last bases up in when the index carries no interval table (4^6 entries: made once per part and seed length class)
__global__ void __launch_bounds__(256)
k_sweep_mini_table(FMView fm, uint32_t q, uint2* __restrict__ table)
{
  __shared__ uint32_t s_sup[SUP_LDS];
  stage_exc_super(fm, s_sup);
  const uint32_t code = blockIdx.x * 256 + threadIdx.x;
  if (code >= (1u << (2 * q))) return;
  uint32_t l = 0, r = fm.n;
  for (uint32_t j = 0; j < q && r > l; ++j) {              // the LAST base first (it sits in the low bits)
    const uint32_t c = (code >> (2 * j)) & 3u;
    const uint32_t nl = fm.C[c] + block_rank(fm, s_sup, fm.blocks + (uint64_t)(l / BLOCK_SYMS) * 4, c, l);
    const uint32_t nr = fm.C[c] + block_rank(fm, s_sup, fm.blocks + (uint64_t)(r / BLOCK_SYMS) * 4, c, r);
    l = nl; r = nr;
  }
  table[code] = r > l ? make_uint2(l, r) : make_uint2(0u, 0u);
}

// one workgroup per bucket
template <bool ROUND0, bool FINAL>
__global__ void __launch_bounds__(256) SW_OCCUPANCY_ATTR
k_fm_sweep(FMView fm, const uint2* __restrict__ table, const uint4* __restrict__ in, const uint64_t* __restrict__ base,
           uint32_t n_staged, uint32_t n_direct, uint32_t gocc_thr, uint4* __restrict__ out,
           uint32_t* __restrict__ iv_lo, uint32_t* __restrict__ iv_cnt, DevCounters* ctr)
{
  __shared__ uint32_t s_sup[SUP_LDS];
  __shared__ uint4 s_blk[SW_CAP * 4];
  __shared__ uint32_t s_cum[SW_CAP * 6];         // per staged block and half-group of 32 symbols: A, C, G, T in front of it inside the block, a byte each
  __shared__ uint32_t s_node_lo[2][SW_NODES], s_node_hi[2][SW_NODES];
  __shared__ uint32_t s_first[SW_NODES], s_base[SW_NODES];
  __shared__ uint32_t s_scan[256 / 64 + 1];
  __shared__ uint32_t s_red[2][4];
  __shared__ uint32_t s_cnt[SW_PER * 4 + 1];
  stage_exc_super(fm, s_sup);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wib = tid >> 6;
  const uint64_t lo0 = base[blockIdx.x], hi0 = base[blockIdx.x + 1];
  uint32_t n_steps = 0, n_live = 0;
  for (uint64_t t0 = lo0; t0 < hi0; t0 += SW_TILE) {
    const uint32_t tn = (uint32_t)min((uint64_t)SW_TILE, hi0 - t0);
    uint32_t l[SW_PER], r[SW_PER], id[SW_PER], kr[SW_PER], nd[SW_PER];
    uint32_t alive = 0;                                     // bit j: seed j of this lane has an interval
    uint4 rec[SW_PER];
    // (every record of a region is a live seed: the scatter that filled the region left the dead ones behind)
#pragma unroll
    for (uint32_t j = 0; j < SW_PER; ++j) {
      const bool in_tile = j * 256 + tid < tn;
      rec[j] = in_tile ? in[t0 + j * 256 + tid] : make_uint4(0, 0, 0, 0);
      if (in_tile) alive |= 1u << j;
    }
    if (ROUND0) {
      uint2 iv[SW_PER];
#pragma unroll
      for (uint32_t j = 0; j < SW_PER; ++j) iv[j] = (alive >> j) & 1u ? table[rec[j].x] : make_uint2(0u, 0u);
#pragma unroll
      for (uint32_t j = 0; j < SW_PER; ++j) { rec[j].x = iv[j].x; rec[j].y = iv[j].y; if (iv[j].y <= iv[j].x) alive &= ~(1u << j); }
    }
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
    for (uint32_t j = 0; j < SW_PER; ++j) {
      l[j] = rec[j].x; r[j] = rec[j].y; id[j] = rec[j].z; kr[j] = rec[j].w; nd[j] = 0;
      if ((alive >> j) & 1u) { mn = min(mn, l[j]); mx = max(mx, r[j]); }
    }
    // the rows the tile's intervals lie in
    for (int d = 32; d > 0; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d)); }
    __syncthreads();                                        // (the previous tile is done with LDS)
    if (lane == 0) { s_red[0][wib] = mn; s_red[1][wib] = mx; }
    __syncthreads();
    mn = min(min(s_red[0][0], s_red[0][1]), min(s_red[0][2], s_red[0][3]));
    mx = max(max(s_red[1][0], s_red[1][1]), max(s_red[1][2], s_red[1][3]));
    const bool any = mx > mn;
    if (tid == 0) { s_node_lo[0][0] = any ? mn : 0u; s_node_hi[0][0] = any ? mx : 0u; }
    uint32_t cur = 0;
    // n_staged steps from LDS (the tree of ranges), then n_direct steps that go to memory (the tail of the seed: what is left
    // after the last round's tree is too little for a round of its own)
    for (uint32_t s = 0; s < n_staged + n_direct && any; ++s) {
      const bool tree = s < n_staged;
      const uint32_t n_nodes = tree ? 1u << (2 * s) : 0u;
      bool staged = false;
      uint32_t nofs = 0;                            // where this wave reads the nodes' first slots (its own copy, or the workgroup's)
      if (tree) {
        __syncthreads();
        // ---- the level's rank blocks: every node's block range, one after the other in the staging area ----
        // Up to 64 nodes (the first four levels): EVERY wave scans the nodes' block counts itself and keeps its own copy of the
        // first slots (s_base / s_first hold four copies of 64 entries) -- nothing crosses a wave, so the two barriers of the
        // scan go (a level is a chain of barriers and one memory round trip: 5 us per tile and level by the counters).  The
        // last level's 256 nodes are scanned by the workgroup.
        const bool own = n_nodes <= 64;
        nofs = own ? wib * 64 : 0u;
        const uint32_t node = own ? lane : tid;
        uint32_t first = 0, nb = 0;
        if (node < n_nodes) {
          const uint32_t a = s_node_lo[cur][node], b = s_node_hi[cur][node];
          if (b > a) { first = a / BLOCK_SYMS; nb = b / BLOCK_SYMS - first + 1; }
        }
        uint32_t incl = nb;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, d); if (lane >= (uint32_t)d) incl += u; }
        uint32_t total, base;
        if (own) {
          total = (uint32_t)__shfl((int)incl, 63);
          base = incl - nb;
          s_first[nofs + lane] = first; s_base[nofs + lane] = base;
          __builtin_amdgcn_s_waitcnt(0xc07f);          // (lgkmcnt(0): the wave's own LDS stores before its own loads)
          __builtin_amdgcn_wave_barrier();
        } else {
          if (lane == 63) s_scan[wib] = incl;
          __syncthreads();
          uint32_t before = 0;
          total = 0;
#pragma unroll
          for (uint32_t w = 0; w < 4; ++w) { if (w < wib) before += s_scan[w]; total += s_scan[w]; }
          base = before + incl - nb;
          if (tid < n_nodes) { s_first[tid] = first; s_base[tid] = base; }
          __syncthreads();
        }
        staged = total != 0 && total <= SW_CAP;      // (no block at all: every interval of the tile has died)
        if (staged) {
          // One lane per block: which node's range the slot lies in (a search over the nodes' first slots: at most 256), the
          // block's 64 bytes by four loads, the counts in front of its six half-groups in registers -- no list of blocks filled by
          // one lane (65-130 stores in a row at the tree's root), no second pass over the counts, two barriers less per level.
          // All of a lane's loads before the first store.
          constexpr uint32_t TURNS = (SW_CAP + 255) / 256;
          uint4 tmp[TURNS][4];
#pragma unroll
          for (uint32_t u = 0; u < TURNS; ++u) {
            const uint32_t slot = min(tid + u * 256, total - 1);      // (past the end: the last block again, dropped below -- no branch round a load)
            uint32_t lo_n = 0, hi_n = n_nodes;                        // the last node whose first slot is <= slot (empty nodes share a first slot
            while (hi_n - lo_n > 1) {                                 // with the node behind them: the LAST of equals is the one that holds blocks)
              const uint32_t mid = (lo_n + hi_n) >> 1;
              if (s_base[nofs + mid] <= slot) lo_n = mid; else hi_n = mid;
            }
            const uint64_t blk = s_first[nofs + lo_n] + (slot - s_base[nofs + lo_n]);
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) tmp[u][w] = fm.blocks[blk * 4 + w];
          }
#pragma unroll
          for (uint32_t u = 0; u < TURNS; ++u) {
            const uint32_t slot = tid + u * 256;
            if (slot < total) {
              uint32_t run = 0;
#pragma unroll
              for (uint32_t w = 0; w < 4; ++w) {
                const uint4 v = tmp[u][w];
                s_blk[slot * 4 + w] = v;
                if (w) {                           // a group of 64 symbols: the characters of its two halves (bytes add without carries: at most 160 each)
                  const uint32_t c0 = __popc(v.x & ~v.z), g0 = __popc(~v.x & v.z), t0 = __popc(v.x & v.z);
                  const uint32_t c1 = __popc(v.y & ~v.w), g1 = __popc(~v.y & v.w), t1 = __popc(v.y & v.w);
                  s_cum[slot * 6 + (w - 1) * 2] = run;
                  run += (32u - c0 - g0 - t0) | (c0 << 8) | (g0 << 16) | (t0 << 24);
                  s_cum[slot * 6 + (w - 1) * 2 + 1] = run;
                  run += (32u - c1 - g1 - t1) | (c1 << 8) | (g1 << 16) | (t1 << 24);
                }
              }
            }
          }
          __syncthreads();
        }
      }
      // ---- every live seed of the tile takes the step ----
      if (staged) {
#pragma unroll
        for (uint32_t j = 0; j < SW_PER; ++j) {
          if (!(alive & (1u << j))) continue;
          const uint32_t c = kr[j] & 3u;
          kr[j] >>= 2;
          uint32_t nl, nr;
          const uint32_t bl = l[j] / BLOCK_SYMS, br = r[j] / BLOCK_SYMS;
          const uint32_t sl = s_base[nofs + nd[j]] + (bl - s_first[nofs + nd[j]]);
          staged_rank_pair(fm, s_sup, s_blk, s_cum, sl, sl + (br - bl), c, l[j], r[j], nl, nr);
          l[j] = fm.C[c] + nl; r[j] = fm.C[c] + nr; nd[j] = (nd[j] * 4 + c) & (SW_NODES - 1);
          ++n_steps;
          if (nr <= nl) alive &= ~(1u << j);
        }
      } else {
        // from memory: the blocks of two seeds at a time, all their loads before the first is looked at
#pragma unroll
        for (uint32_t half = 0; half < SW_PER; half += 2) {
          uint4 a0, a1, a2, a3, b0, b1, b2, b3;
          {
            const uint64_t ba = (alive >> half) & 1u ? l[half] / BLOCK_SYMS : 0u, bb = (alive >> (half + 1)) & 1u ? l[half + 1] / BLOCK_SYMS : 0u;
            a0 = fm.blocks[ba * 4]; a1 = fm.blocks[ba * 4 + 1]; a2 = fm.blocks[ba * 4 + 2]; a3 = fm.blocks[ba * 4 + 3];
            b0 = fm.blocks[bb * 4]; b1 = fm.blocks[bb * 4 + 1]; b2 = fm.blocks[bb * 4 + 2]; b3 = fm.blocks[bb * 4 + 3];
          }
#pragma unroll
          for (uint32_t jj = 0; jj < 2; ++jj) {
            const uint32_t j = half + jj;
            if (!(alive & (1u << j))) continue;
            const uint4 blk[4] = { jj ? b0 : a0, jj ? b1 : a1, jj ? b2 : a2, jj ? b3 : a3 };
            const uint32_t c = kr[j] & 3u;
            kr[j] >>= 2;
            uint32_t nl, nr;
            const uint32_t bl = l[j] / BLOCK_SYMS, br = r[j] / BLOCK_SYMS;
            if (br == bl) block_rank_pair(fm, s_sup, &blk[0], &blk[0], true, c, l[j], r[j], nl, nr);
            else { nl = block_rank(fm, s_sup, &blk[0], c, l[j]); nr = block_rank(fm, s_sup, fm.blocks + (uint64_t)br * 4, c, r[j]); }
            l[j] = fm.C[c] + nl; r[j] = fm.C[c] + nr; nd[j] = (nd[j] * 4 + c) & (SW_NODES - 1);
            ++n_steps;
            if (nr <= nl) alive &= ~(1u << j);
          }
        }
      }
      // ---- the images of every node's range: the next level's nodes ----
      if (s + 1 < n_staged) {
        for (uint32_t ch = tid; ch < n_nodes * 4; ch += 256) {
          const uint32_t p = ch >> 2, c = ch & 3u;
          const uint32_t a = s_node_lo[cur][p], b = s_node_hi[cur][p];
          uint32_t ca = 0, cb = 0;
          if (b > a) {
            const uint32_t ba = a / BLOCK_SYMS, bb = b / BLOCK_SYMS;
            if (staged) {
              const uint32_t sl = s_base[nofs + p];      // (a lies in the node's first block)
              staged_rank_pair(fm, s_sup, s_blk, s_cum, sl, sl + (bb - ba), c, a, b, ca, cb);
              ca += fm.C[c]; cb += fm.C[c];
            } else {
              ca = fm.C[c] + block_rank(fm, s_sup, fm.blocks + (uint64_t)ba * 4, c, a);
              cb = fm.C[c] + block_rank(fm, s_sup, fm.blocks + (uint64_t)bb * 4, c, b);
            }
          }
          s_node_lo[cur ^ 1][ch] = ca; s_node_hi[cur ^ 1][ch] = cb > ca ? cb : ca;
        }
        cur ^= 1;
      }
    }
    if (FINAL) {
      // (first row, occurrences) per seed; seeds above the gocc threshold are dropped here (index_iter.hpp:843-847); iv_cnt was
      // zeroed for everybody
#pragma unroll
      for (uint32_t j = 0; j < SW_PER; ++j) {
        if (!(alive & (1u << j))) continue;
        const uint32_t cnt = r[j] - l[j];
        if (cnt > gocc_thr) continue;
        iv_lo[id[j]] = l[j];
        iv_cnt[id[j]] = cnt;
        ++n_live;
      }
    } else {
      // ---- live intervals to the front of the tile's piece of the region: wave ballot + prefix ----
      __syncthreads();
#pragma unroll
      for (uint32_t j = 0; j < SW_PER; ++j) {
        const uint64_t bal = __ballot((alive >> j) & 1u);
        if (lane == 0) s_cnt[j * 4 + wib] = (uint32_t)__popcll(bal);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < SW_PER * 4; ++i) { const uint32_t v = s_cnt[i]; s_cnt[i] = run; run += v; }
        s_cnt[SW_PER * 4] = run;
      }
      __syncthreads();
      const uint32_t n_alive = s_cnt[SW_PER * 4];
#pragma unroll
      for (uint32_t j = 0; j < SW_PER; ++j) {
        const bool a = (alive >> j) & 1u;
        const uint64_t bal = __ballot(a);
        if (a) out[t0 + s_cnt[j * 4 + wib] + (uint32_t)__popcll(bal & lanemask_lt())] = make_uint4(l[j], r[j], id[j], kr[j]);
      }
      for (uint32_t i = n_alive + tid; i < tn; i += 256) out[t0 + i] = make_uint4(SW_DEAD, 0, 0, 0);
    }
  }
  for (int d = 32; d > 0; d >>= 1) { n_steps += __shfl_down(n_steps, d); n_live += __shfl_down(n_live, d); }
  if (lane == 0) {
    if (n_steps) ctr->n_lf_steps.add((unsigned long long)n_steps);
    if (FINAL && n_live) ctr->n_live.add((unsigned long long)n_live);
  }
}

// per-wave totals of the seeds' occurrence counts, in seed order: what k_fm_search leaves for k_wave_offsets
__global__ void __launch_bounds__(256)
k_sweep_totals(const uint32_t* __restrict__ iv_cnt, const uint64_t* __restrict__ params, uint64_t seeds_cap, uint32_t per_wave,
               uint64_t* __restrict__ wave_total)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  uint64_t sum = 0;
  for (uint64_t s = s0 + lane; s < s1; s += 64) sum += iv_cnt[s];
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
  if (lane == 0) wave_total[wave] = sum;
}
