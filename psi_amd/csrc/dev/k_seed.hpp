// K0: seeding (scan, k_seed_pack) and the per-chunk seed table of the traverser -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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
      for (uint32_t w = 0; w < NW; ++w) x[j][w] = 0;
      // (all words of the seed in ONE block, from addresses pulled back inside the buffer where the seed ends near its end --
      // such a seed takes the byte loop below: a load per branch was a memory latency per word, round 5)
      if (n_bases >= 8) {
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w) {
          uint64_t a = (in[j] ? abs0 : 0ull) + 8 * w;
          a = a + 8 <= n_bases ? a : n_bases - 8;
          __builtin_memcpy(&x[j][w], bases + a, 8);
        }
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
  uint32_t tile;             // seeds per workgroup of those kernels (SB_TILE; four times that with the finer partition of large chunks,
                             // so that the count matrix -- buckets x workgroups -- stays a fraction of the data)
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
  const uint64_t s0 = (uint64_t)blockIdx.x * sb.tile, s1 = min(n_seeds, s0 + sb.tile);
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
             const uint64_t* __restrict__ base /* first record of every bucket */, const uint32_t* __restrict__ within /* [bucket][wg]: the
             workgroup's first record inside the bucket (k_sweep_rows) */, ulonglong2* __restrict__ out_rec /* (k-mer, seed) */,
             uint32_t* __restrict__ seed_next)
{
  extern __shared__ uint32_t cur[];
  for (uint32_t i = threadIdx.x; i < sb.n_buckets; i += 256) cur[i] = (uint32_t)base[i] + within[(uint64_t)i * sb.n_wg + blockIdx.x];
  __syncthreads();
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = (uint64_t)blockIdx.x * sb.tile, s1 = min(n_seeds, s0 + sb.tile);
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

