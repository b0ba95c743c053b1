// K4: the traverser -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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
  uint32_t spr = 0, step = 0;                     // equal read lengths (spr seeds per read): a seed's read and offset are its number divided, seed_info is not read
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
      uint2 si;
      if (tb.spr) { const uint32_t rd = s / tb.spr; si = make_uint2(rd, (s - rd * tb.spr) * tb.step); }      // (one random request less per hit)
      else si = tb.seed_info[s];
      rid = rec_offset + si.x; roff = si.y;
      nx = dup;                                   // then down the duplicate chain
      if (dup != NIL) dup = tb.seed_next[dup];
    }
    chunk_emit(cw, has, nid, noff, rid, roff, ctr);
    s = nx;
  }
}

// The prefix walks of the starting loci against the chunk's long prefix map (4^min(k,14) bits): a stream (16 bytes per walk
// in, the few that pass out).  A wave writes what passes of the walks it looked at to a segment of `out` of its own, its count to seg_cnt[wave]: wave w of the
// traverser takes segment w.  (Until round 5 the survivors were appended to one array through one counter:
// 24 576 waves x at least one atomic on one address, ~11 ns each = 0.27 of the kernel's 0.34 ms.)
constexpr uint32_t PF_R = 8;                     // eight rounds of 64 walks per wave and latency chain
constexpr uint32_t PF_CHUNK = 4 * 64 * PF_R;     // walks a workgroup takes at a time
__global__ void __launch_bounds__(256)
k_pfx_filter(const uint4* __restrict__ roots, uint64_t n, uint32_t per_wave, const uint32_t* __restrict__ pfx12,
             const uint32_t* __restrict__ pfx_bits, uint32_t depth, uint4* __restrict__ out, uint32_t* __restrict__ seg_cnt)
{
  // The workgroups sweep the walks as one front -- workgroup b takes chunks b, b + grid, b + 2 grid ... of PF_CHUNK walks, its
  // four waves interleaved round by round -- so that what is in flight at any time is one contiguous window of memory (a wave
  // with a 12-KB range of its own, 24 576 ranges open at once, ran at 1.6 TB/s).  The walks come ordered by prefix, so a
  // round's look-ups are neighbours in the map.  The loads of all PF_R rounds are issued before the first is looked at
  // (walk, then map word: two latencies in a row).  Output: segment `wave` of per_wave slots.
  const uint32_t lane = lane_id(), wib = threadIdx.x >> 6;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + wib;
  uint4* __restrict__ mine = out + wave * per_wave;
  uint32_t held = 0;                              // wave-uniform: walks of this wave that passed
  for (uint64_t base = (uint64_t)blockIdx.x * PF_CHUNK; base < n; base += (uint64_t)gridDim.x * PF_CHUNK) {
    uint4 e[PF_R];
    uint32_t w14[PF_R];
    bool in[PF_R];
#pragma unroll
    for (uint32_t r = 0; r < PF_R; ++r) {
      const uint64_t i = base + (uint64_t)(r * 4 + wib) * 64 + lane;
      in[r] = i < n;
      e[r] = roots[in[r] ? i : n - 1];            // (a lane past the end repeats the last walk and drops the answer)
    }
    // (whole 16-byte loads: left alone the compiler fetches the prefix word now and the other three words later, behind the
    // branch that keeps a walk -- one more memory latency per round, eight in a row per iteration)
#pragma unroll
    for (uint32_t r = 0; r < PF_R; ++r) keep_whole(e[r]);
    // The long map alone, for every walk: the walks are in prefix order, so the 64 look-ups of a round fall into two or
    // three neighbouring sectors of it and the kernel reads the map once, front to back (32 MiB beside 282 MB of walks).
    // Asking the 2-MiB 12-mer map first -- as the traverser did when the walks came in locus order and a look-up in the
    // long map was a random sector -- was a third memory latency in the chain for nothing.
#pragma unroll
    for (uint32_t r = 0; r < PF_R; ++r) w14[r] = pfx_bits[e[r].x >> 5];
    bool pass[PF_R];
#pragma unroll
    for (uint32_t r = 0; r < PF_R; ++r) pass[r] = in[r];
#pragma unroll
    for (uint32_t r = 0; r < PF_R; ++r) {
      const bool keep = pass[r] && ((w14[r] >> (e[r].x & 31)) & 1u);
      const uint64_t km = __ballot(keep);
      if (keep) mine[held + (uint32_t)__popcll(km & lanemask_lt())] = e[r];
      held += (uint32_t)__popcll(km);
    }
  }
  if (lane == 0) seg_cnt[wave] = held;
}

template <bool ENUM, typename KEY = uint64_t, bool WINDOW = true>      // (WINDOW false: roots that are not in node order -- the filtered
__global__ void __launch_bounds__(64)                                 // prefix walks; 4 KB of LDS less per wave = 30 waves per CU, not 17)
k_traverse(GraphView g, TableView tb, const uint2* __restrict__ loci /* (node rank, offset) */,
           uint64_t n_loci, uint32_t loci_per_wave,
           const TravItemT<KEY>* __restrict__ spill_in, uint64_t n_spill_in,
           TravItemT<KEY>* __restrict__ spill_out, uint64_t spill_cap,
           uint32_t k, uint64_t rec_offset, psigpu_hit* __restrict__ chunks, uint32_t* __restrict__ chunk_fill,
           uint32_t cap_chunks, uint64_t n_nodes, DevCounters* ctr, EnumOut eo, const uint4* __restrict__ pfx_roots = nullptr,
           const uint32_t* __restrict__ seg_cnt = nullptr /* pfx_roots in segments of loci_per_wave, the first seg_cnt[wave] of each
                                                             filled (k_pfx_filter): wave w takes segment w */)
{
  // pfx_roots (query time, k > 12): the roots are not the loci but their PREFIX WALKS, enumerated once per index
  // (ensure_pfx_roots): (prefix, node, offset in the node's record, locus) -- where a walk from the locus stands after
  // min(k, 14) bases -- and of those only the ones that pass the chunk's two prefix maps (k_pfx_filter, round 5: a streaming
  // kernel; in round 4 this kernel staged all of them and tested the 12-mer map itself, one memory latency per 64 roots
  // and wave, and walked the survivors to 14 bases through the graph).  Every lane starts on a walk that matters.
  typedef TravItemT<KEY> TravItem;
  typedef DoneItemT<KEY> DoneItem;
  static_assert(!ENUM || sizeof(KEY) == 8, "the tables are made for one-word seeds");
  __shared__ TravItem stack[TRAV_CAP];
  __shared__ DoneItem doneq[DONE_CAP];
  __shared__ TravItem rootbuf[64];        // staged roots and their start offsets
  __shared__ uint32_t rootoff[64];
  __shared__ NodeLite window[WINDOW ? TRAV_WIN : 1];   // node records of the ranks this wave's loci start in
  ChunkWriter cw = { chunks, chunk_fill, cap_chunks, NIL, 0 };
  PairWriter pw = { NIL, 0 };
  const uint32_t lane = lane_id();
  // roots: either fresh loci (spill_in == nullptr) or spilled partial walks
  const bool from_spill = spill_in != nullptr;
  const bool from_pfx = !ENUM && !from_spill && pfx_roots != nullptr;      // (n_loci then counts prefix walks)
  uint64_t n_roots = from_spill ? n_spill_in : n_loci;
  uint64_t cursor = min(n_roots, (uint64_t)blockIdx.x * loci_per_wave);     // next root NOT yet requested from memory
  const uint64_t cend = (from_pfx && seg_cnt) ? cursor + min(seg_cnt[blockIdx.x], loci_per_wave) : min(n_roots, cursor + loci_per_wave);
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
        const uint4 e = pfx_roots[rix];       // prefix of tb.pfx_len bases, node, offset, locus (both prefix maps passed)
        pf.kmer = (KEY)e.x | ((KEY)1 << (2 * tb.pfx_len)); pf.node = e.y; pf.locus = e.w; pf_off = e.z;
      }
      else { uint2 lc = loci[rix]; pf.kmer = 1; pf.node = lc.x; pf.locus = (uint32_t)rix; pf_off = lc.y; }
    }
    cursor += pf_cnt;
  };
  // The loci of a wave are consecutive, so are the ranks of the nodes they start in, and (for
  // graphs whose ranks follow the topology, as vg's do) so are the nodes the walks hop to: stage
  // that rank window in LDS once, coalesced; anything outside is read from memory.
  uint32_t wb = 0, win_n = 0;             // first rank / size of the window (none for spill launches)
  if (WINDOW && !from_spill && !from_pfx && cursor < cend) {      // (prefix walks come ordered by prefix: their nodes are anywhere)
    wb = loci[cursor].x;
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
      if (WINDOW && widx < win_n) nlw = *reinterpret_cast<const uint4*>(&window[widx]); else nlw = *reinterpret_cast<const uint4*>(&g.lite[it.node]);
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

