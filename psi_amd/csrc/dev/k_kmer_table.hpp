// the k-mer table of the default mode: slots, construction, probe; row records of the FM modes -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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
  // (KP_R rounds per iteration, round 5: the keys of all rounds, then their slots, each group in one block -- k_kmer_step's
  // lesson; one round at a time this kernel had one probe per lane in flight)
  constexpr uint32_t KP_R = 4;
  for (uint64_t base0 = s0; base0 < s1; base0 += 64 * KP_R) {
    uint64_t keys[KP_R], hs[KP_R];
    uint4 vs[KP_R];
#pragma unroll
    for (uint32_t rr = 0; rr < KP_R; ++rr) {
      const uint64_t sd = base0 + 64 * rr + lane;
      keys[rr] = seed_key[sd < s1 ? sd : s0];
    }
#pragma unroll
    for (uint32_t rr = 0; rr < KP_R; ++rr) {
      if (base0 + 64 * rr + lane >= s1) keys[rr] = KEY_INVALID;
      hs[rr] = keys[rr] != KEY_INVALID ? kt_home(keys[rr], kt.n_slots) : 0ull;      // (a seed with an N loads slot 0: no branch between the loads)
      vs[rr] = *reinterpret_cast<const uint4*>(kt.ht + hs[rr]);
    }
#pragma unroll
    for (uint32_t rr = 0; rr < KP_R; ++rr) {
    const uint64_t seed = base0 + 64 * rr + lane;
    if (seed >= s1) continue;
    const uint64_t key = keys[rr];
    keep_whole(vs[rr]);
    uint4 res = make_uint4(0, 0, 0, 0);
    if (key != KEY_INVALID) res = kt_resolve(kt, key, hs[rr], vs[rr], want_on, want_off, gocc_thr);
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
  // FD_R rounds of 64 seeds per iteration (round 5): the keys of all rounds, then their interval-table entries and table
  // probes, then the second rows' records are each issued together -- one round at a time the kernel spent a memory
  // latency per load and round (key -> entry -> row), with one seed per lane in flight.
  constexpr uint32_t FD_R = 4;
  const bool have_lk = lk.ht != nullptr;
  for (uint64_t base = s0; base < s1; base += 64 * FD_R) {
    uint64_t key[FD_R];
    bool in[FD_R];
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) {
      const uint64_t seed = base + 64 * rr + lane;
      in[rr] = seed < s1;
      key[rr] = seed_key[in[rr] ? seed : s0];      // (a lane past the end reads the wave's first key and drops the answer)
    }
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) if (!in[rr]) key[rr] = KEY_INVALID;
    uint64_t h[FD_R];
    uint4 slw[FD_R], ea[FD_R];
    uint2 eb[FD_R];
    // (no branch around a load, not even a wave-uniform one: a block that ends behind a load ends in a wait for it.  A finder
    // without a locus table probes the key array instead, and drops what it gets)
    const uint4* const lk_base = have_lk ? reinterpret_cast<const uint4*>(lk.ht) : reinterpret_cast<const uint4*>(seed_key);
    const char* const ft_base = ftabx ? reinterpret_cast<const char*>(ftabx) : reinterpret_cast<const char*>(fm.ftab);
    const uint32_t ft_shift = ftabx ? 5u : 3u;      // 32-byte entries with the first row's record, or 8-byte intervals
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) {
      const bool valid = key[rr] != KEY_INVALID;
      h[rr] = (have_lk && valid) ? lkt_home(key[rr], lk.n_slots) : 0ull;
      slw[rr] = lk_base[h[rr]];
      const char* e = ft_base + ((valid ? (key[rr] & qmask) : 0ull) << ft_shift);
      __builtin_memcpy(&ea[rr], e, 16);             // l, r, first row's node and offset (or l, r and the next entry's)
      __builtin_memcpy(&eb[rr], e + 16, 8);         // its context (or the entry after that: dropped)
    }
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) {
      if (!have_lk) slw[rr] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
      if (!ftabx) { ea[rr].z = ea[rr].w = 0; eb[rr] = make_uint2(0, 0); }
    }
    // the second row of every interval that has one (1.45 rows on average): its record, before any round is looked at
    uint4 row1[FD_R];
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) {
      const bool valid = key[rr] != KEY_INVALID;
      const uint32_t l_ = ea[rr].x, r_ = ea[rr].y;
      const uint32_t cnt_ = (valid && r_ > l_) ? r_ - l_ : 0u;
      const bool want1 = rem != 0 && cnt_ >= 2 && cnt_ <= VERIFY_ROWS;
      row1[rr] = *reinterpret_cast<const uint4*>(&fm.sarec[want1 ? l_ + 1 : 0u]);
    }
#pragma unroll
    for (uint32_t rr = 0; rr < FD_R; ++rr) {
      const uint64_t seed = base + 64 * rr + lane;
      const bool valid = key[rr] != KEY_INVALID;
      const uint32_t l = valid ? ea[rr].x : 0u, r = valid ? ea[rr].y : 0u;
      SaRec first = { ea[rr].z, ea[rr].w, (uint64_t)eb[rr].x | ((uint64_t)eb[rr].y << 32) };      // row l's record, when the interval table carries it
      uint32_t cnt = r > l ? r - l : 0u, aux = 0, on_node = 0, on_noff = 0;
      const bool deferred = rem != 0 && cnt > VERIFY_ROWS;
      if (rem != 0 && cnt != 0 && !deferred) {
        // the rem bases in front of each row against the head of the seed, rows four at a time; the
        // first matching row's record also gives K2 the hit itself
        const uint64_t want = key[rr] >> (2 * q);
        uint32_t mask = 0;
        for (uint32_t t0 = 0; t0 < cnt; t0 += 4) {
          SaRec c[4];
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j) {
            c[j] = SaRec{ 0, 0, 0 };
            if (t0 + j < cnt) {
              if (ftabx && t0 + j == 0) { c[j] = first; continue; }
              uint4 v = row1[rr];
              if (t0 + j != 1) v = *reinterpret_cast<const uint4*>(&fm.sarec[l + t0 + j]);
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
      if (have_lk && valid) {
        const TableSlot sl = { (unsigned long long)slw[rr].x | ((unsigned long long)slw[rr].y << 32), slw[rr].z, slw[rr].w };
        lkt_resolve(lk, key[rr], h[rr], sl, ofirst, ocnt, onoff);
      }
      const bool keep = !deferred && cnt != 0 && cnt <= gocc_thr;
      if (in[rr]) {
        so.iv_lo[seed] = l;
        so.iv_cnt[seed] = keep ? cnt : 0u;
        so.iv_aux[seed] = aux;
        so.on_node[seed] = on_node;
        so.on_noff[seed] = on_noff;
        if (have_lk) { so.off_first[seed] = ofirst; so.off_cnt[seed] = ocnt; so.off_noff[seed] = onoff; }
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

