// the default step in one kernel -- part of the one translation unit device.hip (included there, in order; not a header of its own).
// ------------------------------------------------------------------------------------
// The default step in ONE kernel (round 5): seeding, the k-mer table probe and the emission of a TILE of seeds by one
// workgroup, where rounds 1-4 ran k_seed_pack -> k_kmer_probe -> k_kmer_emit with 8 bytes of key and 8 bytes of result per
// seed written and read back in between (a third of the step's traffic, two launches, and a probe kernel with one load
// in flight per lane).  A workgroup takes the next tile (KS_R rounds of 256 seeds; a ticket, so tiles start in order),
// packs its seeds' keys in registers, issues the first table load of all its rounds before it looks at any, counts the
// tile's hits and learns its first output slot by a decoupled look-back over the tiles before it, in two levels (tiles of
// its group of 64, groups before its group): tile_state[t] / group_state[g] is ONE 64-bit word -- flag (aggregate / inclusive
// prefix), the call's serial number, the count -- so a word is either this call's or ignored and nothing needs a fence.  The records come out in seed order exactly as k_kmer_emit writes them
// (emit_round: the transposed stores, the spread of a seed with many hits over the wave).
// The general (not equal-length) reads locate their read as k_seed_pack does, from the scanned seed offsets.
// Traverse mode runs the same kernel over the table of the PATHS' k-mers and has it leave the seeds' k-mers and (read, offset)
// behind for the chunk's seed table and the traverser.
// ------------------------------------------------------------------------------------
#ifndef KS_ROUNDS
#define KS_ROUNDS 4
#endif
constexpr int KS_R = KS_ROUNDS;      // (measured: tools/r05_step_ab.sh)
constexpr uint32_t KS_TILE = 256 * KS_R;
constexpr uint64_t KS_AGG = 1ull << 62, KS_PFX = 2ull << 62, KS_VAL = (1ull << 40) - 1;
constexpr uint32_t KS_SERIAL = (1u << 22) - 1;
// look-back words a call of up to n seeds needs: one per tile (+ slack), one per group of 64 tiles behind them
__host__ __device__ inline uint64_t ks_state_words(uint64_t n_seeds) { return (n_seeds / KS_TILE + 66) + (n_seeds / KS_TILE) / 64 + 4; }
__device__ __forceinline__ uint64_t ks_word(uint64_t flag, uint32_t serial22, uint64_t v) { return flag | ((uint64_t)serial22 << 40) | v; }

#ifndef KS_OCC_ATTR
#define KS_OCC_ATTR
#endif
template <bool PACKED, bool UNIFORM>
__global__ void __launch_bounds__(256) KS_OCC_ATTR
k_kmer_step(const char* __restrict__ bases, const uint64_t* __restrict__ read_off, const uint64_t* __restrict__ seed_off, uint64_t n_reads,
            const uint64_t* __restrict__ params, uint64_t seeds_cap, uint64_t n_bases, uint32_t k, uint32_t step, PackedIn pk, UniformIn un,
            KmerTableView kt, MapView mv, const LocusEnt* __restrict__ ent, bool want_on, bool want_off, uint32_t gocc_thr,
            uint64_t rec_offset, psigpu_hit* __restrict__ hits, uint64_t cap, uint64_t* tile_state, uint32_t serial22,
            DevCounters* ctr, uint32_t opts /* 4: PSIGPU_ANY_ORDER -- the tile's output range by one atomic add */,
            uint64_t* __restrict__ seed_key_out = nullptr, uint2* __restrict__ seed_info_out = nullptr /* traverse mode: the seeds'
                                               k-mers and (read, offset) for the chunk's seed table and the traverser behind this kernel */)
{
  __shared__ uint32_t s_tile;
  __shared__ uint32_t s_cnt[KS_R * 4];
  __shared__ uint64_t s_prefix;
  const uint32_t lane = lane_id(), wib = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = (uint32_t)atomicAdd(&ctr->ticket.v, 1ull);
  __syncthreads();
  const uint64_t tile = s_tile;
  // (equal lengths: the seed count is the launch's own -- no load between the ticket and the first address)
  const uint64_t n_seeds = UNIFORM ? seeds_cap : min(params[0], seeds_cap);
  const uint64_t t0 = tile * KS_TILE;
  if (t0 >= n_seeds) return;                      // (the grid is sized by the upper bound of the seed count)

  // A workgroup lives for a handful of memory latencies in a row -- ticket, bases, table slot, look-back -- and a round's
  // loads were first written inside the round's own branch, used before the next round issued its own: KS_R latencies per
  // phase instead of one (63 us per workgroup).  Every phase now ISSUES the loads of all its rounds, then consumes them.
  // ---- the seeds of this lane: read and offset in the read ---------------------------------------------------------------
  bool have[KS_R];
  uint64_t abs0[KS_R];
  uint2 si[KS_R];
  uint64_t sd[KS_R];
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    const uint64_t s_raw = t0 + (uint64_t)r * 256 + threadIdx.x;
    have[r] = s_raw < n_seeds;
    sd[r] = have[r] ? s_raw : n_seeds - 1;        // (a lane past the end repeats the last seed's addresses and drops the result)
  }
  if constexpr (UNIFORM) {
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint32_t rd = (uint32_t)sd[r] / un.spr;
      const uint32_t st = ((uint32_t)sd[r] - rd * un.spr) * step;
      si[r] = make_uint2(rd, st);
      abs0[r] = (uint64_t)rd * un.len + st;
    }
  } else {
    // the read of a seed: the last one whose scanned seed offset is <= the seed's number (k_seed_pack) -- the proportional
    // guess of every round first (loads that do not depend on each other), a wrong guess gallops / bisects
    const uint64_t ratio = params[1];
    uint64_t rd[KS_R], so0[KS_R], so1[KS_R];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      rd[r] = __umul64hi(sd[r], ratio);
      if (rd[r] >= n_reads) rd[r] = n_reads - 1;
      so0[r] = seed_off[rd[r]]; so1[r] = seed_off[rd[r] + 1];
    }
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint64_t s = sd[r];
      if (!(so0[r] <= s && s < so1[r])) {
        uint64_t l = rd[r], hi;
        if (so0[r] <= s) {
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
        rd[r] = l; so0[r] = seed_off[l];
      }
    }
    uint64_t ro[KS_R];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) ro[r] = read_off[rd[r]];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint64_t st = (sd[r] - so0[r]) * step;
      si[r] = make_uint2((uint32_t)rd[r], (uint32_t)st);
      abs0[r] = ro[r] + st;
    }
  }
  // ---- their bases: all loads of all rounds, then the keys ------------------------------------------------------------------
  uint64_t key[KS_R];
  uint32_t nok = 0;
  if constexpr (PACKED) {
    const uint64_t* __restrict__ P = reinterpret_cast<const uint64_t*>(bases);
    uint64_t w0[KS_R], w1[KS_R], m0[KS_R], m1[KS_R];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint64_t w = (abs0[r] + pk.bias2) >> 5;
      w0[r] = P[w]; w1[r] = P[w + 1];             // (the buffer is padded: the word behind the window is loaded, none of its bits used)
      // (no mask: the same two words again instead of a branch around the loads -- a load inside a branch is waited for at
      // the branch's end, KS_R latencies in a row; the result is dropped below)
      const uint64_t* __restrict__ M = pk.mask ? pk.mask : P;
      const uint64_t qm = pk.mask ? (abs0[r] + pk.biasm) >> 6 : w;
      m0[r] = M[qm]; m1[r] = M[qm + 1];
    }
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint32_t sh = 2u * (uint32_t)((abs0[r] + pk.bias2) & 31);
      const uint64_t kk = (sh ? (w0[r] << sh) | (w1[r] >> (64 - sh)) : w0[r]) >> (64 - 2 * k);
      const uint32_t ms = (uint32_t)((abs0[r] + pk.biasm) & 63);
      const uint64_t win = pk.mask ? (ms ? (m0[r] >> ms) | (m1[r] << (64 - ms)) : m0[r]) : 0ull;
      const bool ok = have[r] && (win & ((1ull << k) - 1ull)) == 0;
      key[r] = ok ? kk : KEY_INVALID;
      nok += ok;
    }
  } else {
    const uint32_t nw = (k + 7) >> 3;
    uint64_t x[KS_R][4];
    bool fast[KS_R];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      fast[r] = abs0[r] + 8ull * nw <= n_bases;
#pragma unroll
      for (uint32_t w = 0; w < 4; ++w) x[r][w] = 0;
    }
    // four words whatever k is, from an address pulled back inside the buffer when the seed ends near its end (such a seed
    // takes the byte loop below and its words are not looked at): sixteen loads in ONE block, no branch between them
    if (n_bases >= 8) {
#pragma unroll
      for (int r = 0; r < KS_R; ++r)
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
          uint64_t a = abs0[r] + 8 * w;
          a = a + 8 <= n_bases ? a : n_bases - 8;
          __builtin_memcpy(&x[r][w], bases + a, 8);
        }
    }
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      uint64_t kk = 0;
      uint32_t ok = 1;
      if (fast[r]) {
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w)
          if (w < nw) {
            const uint32_t take = min(8u, k - 8 * w);
            kk = (kk << (2 * take)) | pack8(x[r][w], take, ok);
          }
      } else {
        const char* p = bases + abs0[r];
        for (uint32_t i = 0; i < k; ++i) {        // tail of the buffer: byte loads
          int b = base2(p[i]);
          if (b < 0) { ok = 0; b = 0; }
          kk = (kk << 2) | (uint64_t)b;
        }
      }
      ok = have[r] ? ok : 0u;
      key[r] = ok ? kk : KEY_INVALID;
      nok += ok;
    }
  }

  if (seed_key_out) {
#pragma unroll
    for (int r = 0; r < KS_R; ++r)
      if (have[r]) { seed_key_out[sd[r]] = key[r]; if (seed_info_out) seed_info_out[sd[r]] = si[r]; }      // (equal lengths: nobody reads them)
  }
  // ---- one probe per seed: every round's first load in flight before the first is looked at ----------------------------
  uint64_t h[KS_R];
  uint4 v[KS_R];
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    // (an invalid seed loads its slot 0 all the same: no branch between the loads)
    h[r] = key[r] != KEY_INVALID ? kt_home(key[r], kt.n_slots) : 0ull;
    v[r] = *reinterpret_cast<const uint4*>(kt.ht + h[r]);
  }
  if constexpr (UNIFORM) {
    // the claim, checked where it is used (and behind the probe loads: nothing waits for it): the reads of this lane's seeds
    // start and end where equal lengths put them
    uint64_t ra[KS_R], rb[KS_R];
#pragma unroll
    for (int r = 0; r < KS_R; ++r) { ra[r] = read_off[si[r].x]; rb[r] = read_off[si[r].x + 1]; }
    uint32_t bad = 0;                             // (no short circuit: a branch per load would put their latencies in a row)
#pragma unroll
    for (int r = 0; r < KS_R; ++r) {
      const uint64_t ro = (uint64_t)si[r].x * un.len;
      bad |= (uint32_t)(ra[r] != ro) | (uint32_t)(rb[r] != ro + un.len);
    }
    if (bad) ctr->not_uniform.v = 1ull;
  }
  uint4 res[KS_R];
  uint32_t on_sum = 0, n_live = 0;
#pragma unroll
  for (int r = 0; r < KS_R; ++r) {
    keep_whole(v[r]);
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
    if (opts & 4u) {
      // any order: the tiles' ranges in the order the tiles get here -- nobody waits for a tile in front (n_hits_tab ends as the total)
      unsigned long long at = 0;
      if (lane == 0) at = atomicAdd(&ctr->n_hits_tab.v, (unsigned long long)agg);
      excl = (uint64_t)__shfl((unsigned long long)at, 0);
    } else
    {
      // Two levels (round 6).  Round 5 walked back over the TILES, 64 per step, until it met a tile that had published its
      // prefix: with ~2 000 tiles of the same age resident, none of the tiles just in front has one yet, and a tile walked up to
      // thirty windows -- thirty memory latencies in a row -- while holding its slot on the CU (0.06 ms of the kernel).  Now
      // tiles come in GROUPS of 64: a tile adds the aggregates of the tiles before it in its own group (one window, one
      // load) and the groups before its group (a second load, issued beside the first: at most ~110 groups, of which the ~30
      // in flight have at least their aggregate out as soon as their 64 tiles have counted).  A group's words are published
      // by its last tile: the aggregate when the group's tiles have all counted, the inclusive prefix when that tile knows
      // its own.  Nobody waits for a prefix: an aggregate is enough to go on.
      const uint64_t grp = tile >> 6;
      const uint32_t pos = (uint32_t)tile & 63u;
      uint64_t* const group_state = tile_state + ((seeds_cap + KS_TILE - 1) / KS_TILE + 1);
      if (lane == 0) __hip_atomic_store(&tile_state[tile], ks_word(KS_AGG, serial22, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t top = (int64_t)grp - 1;               // the window over the groups: groups top, top - 1, ... on lanes 0 .. 63
      auto load_groups = [&]() -> uint64_t {
        const int64_t idx = top - (int64_t)lane;
        return idx >= 0 ? __hip_atomic_load(&group_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ks_word(KS_PFX, serial22, 0);
      };
      uint64_t wg = load_groups();                  // (in flight beside the first look at the own group)
      uint64_t s_in = 0;
      if (pos) {
        while (true) {
          const uint64_t w = lane < pos ? __hip_atomic_load(&tile_state[(grp << 6) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
          const bool ready = lane >= pos || ((w >> 62) != 0 && (uint32_t)((w >> 40) & KS_SERIAL) == serial22);
          if (__ballot(ready) == ~0ull) {
            uint64_t part = lane < pos ? (w & KS_VAL) : 0ull;
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
            s_in = part;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const bool closes = pos == 63u;
      if (closes && lane == 0) __hip_atomic_store(&group_state[grp], ks_word(KS_AGG, serial22, s_in + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool fresh = true;
      while (true) {
        const uint64_t w = fresh ? wg : load_groups();
        fresh = false;
        const bool ready = (w >> 62) != 0 && (uint32_t)((w >> 40) & KS_SERIAL) == serial22;
        const uint64_t m_ready = __ballot(ready), m_pfx = __ballot(ready && (w >> 62) == 2);
        const uint32_t n_ready = m_ready == ~0ull ? 64u : (uint32_t)__ffsll((long long)~m_ready) - 1u;      // groups ready from the window's top
        const uint32_t first_pfx = m_pfx ? (uint32_t)__ffsll((long long)m_pfx) - 1u : 64u;
        if (first_pfx < n_ready || (first_pfx == 64u && n_ready == 64u)) {
          const uint32_t upto = first_pfx < 64u ? first_pfx : 63u;      // add lanes 0 .. upto
          uint64_t part = lane <= upto ? (w & KS_VAL) : 0ull;
          for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
          excl += part;
          if (first_pfx < 64u) break;
          top -= 64;
        } else __builtin_amdgcn_s_sleep(1);
      }
      excl += s_in;
      if (closes && lane == 0) __hip_atomic_store(&group_state[grp], ks_word(KS_PFX, serial22, excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      s_prefix = excl;
      if (!(opts & 4u) && t0 + KS_TILE >= n_seeds) ctr->n_hits_tab.v = excl + agg;    // the last tile: what the step wrote (or would have, past cap)
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
    emit_round(mv, ent, sh, cnt, si[r], mine, rec_offset, hits, cap);
    woff += (uint64_t)s_cnt[r * 4] + s_cnt[r * 4 + 1] + s_cnt[r * 4 + 2] + s_cnt[r * 4 + 3];
  }
}

