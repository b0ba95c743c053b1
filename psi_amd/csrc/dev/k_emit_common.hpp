// K2: what the locate / emit kernels read of the index, output writers -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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


