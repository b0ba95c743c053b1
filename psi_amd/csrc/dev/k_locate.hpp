// K2: locate + map + emit (sampled and whole suffix array, k-mer table mode) -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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
            uint32_t uni_spr = 0, uint32_t uni_step = 0 /* seed_info == nullptr: seed s is seed s % spr of read s / spr */)
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
      emit_round(mv, ent, sh, cnt, si, woff, rec_offset, hits, cap);
    }
  }
}

