// MEM mode kernels -- part of the one translation unit device.hip (included there, in order; not a header of its own).
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

