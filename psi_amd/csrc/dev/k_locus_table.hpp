// the locus k-mer table: construction kernels and probe -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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
    keys[dst0 + i] = (uint32_t)r.x;              // ordered by the PREFIX (round 5; by locus in round 4): k_pfx_filter then reads the
    vals[dst0 + i] = (uint32_t)(dst0 + i);       // chunk's prefix maps front to back instead of at 34 M random words per chunk
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

