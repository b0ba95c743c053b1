// K1: FM backward search (LF steps over the rank blocks), quad per seed -- part of the one translation unit device.hip (included there, in order; not a header of its own).
// ------------------------------------------------------------------------------------
// K1: FM backward search, one quad per seed.  The text is the FORWARD path text, so the
// seed is consumed from its last base to its first (the reference appends characters to a
// pattern on the REVERSED text: index_iter.hpp:820-824 -- same occurrences).
// Interval [l, r) half-open; l' = C[c] + rank_c(l), r' = C[c] + rank_c(r)
// (sdsl::backward_search behind fmindex.hpp:856).
// ------------------------------------------------------------------------------------
// Every wave owns a contiguous range of `per_wave` seeds (a multiple of 16) and walks it 16
// seeds -- one per quad -- at a time.  It leaves (lo, count) per seed and the sum of its counts;
// k_wave_offsets turns the sums into the wave's first output slot, and k_fm_locate, walking the
// same ranges, places every hit with a running wave-local prefix: hits come out in seed order
// with no atomics and no scan over the seeds.
template <bool LISTED, typename KEY>   // (LISTED: two kernels -- the list mode's pointers and strides cost the range mode scalar
                              // registers, and at 101 of them a SIMD holds 7 waves instead of the 8 the launch is sized
                              // for.  KEY: one word for seeds of up to 31 bases, two for up to 63)
__global__ void __launch_bounds__(256)
k_fm_search(FMView fm, const KEY* __restrict__ seed_key, const uint64_t* __restrict__ params,
            uint64_t seeds_cap, uint32_t per_wave,
            uint32_t k, uint32_t gocc_thr, uint32_t* __restrict__ iv_lo, uint32_t* __restrict__ iv_cnt,
            uint32_t* __restrict__ iv_aux, uint64_t* __restrict__ wave_total, DevCounters* ctr,
            const uint32_t* __restrict__ list, const unsigned long long* __restrict__ n_list)
{
  // Two ways to be given work: every wave owns the contiguous seed range [wave * per_wave, ...)
  // (list == nullptr), or the waves share a list of seed indices -- the seeds k_fm_search_direct
  // deferred -- and add each seed's count to the total of the wave that owns its range.
  __shared__ uint32_t s_sup[SUP_LDS];
  stage_exc_super(fm, s_sup);
  const bool can_verify = fm.text4 != nullptr && fm.sa != nullptr;
  constexpr bool listed = LISTED;
  const uint32_t ql = threadIdx.x & 3, quad = (threadIdx.x & 63) >> 2;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = listed ? wave * 16 : wave * per_wave;
  const uint64_t s1 = listed ? min((uint64_t)*n_list, seeds_cap) : min(n_seeds, s0 + per_wave);
  const uint64_t stride = listed ? ((uint64_t)gridDim.x * blockDim.x >> 6) * 16 : 16;
  uint32_t n_live = 0, n_steps = 0, n_rows = 0;
  uint64_t wsum = 0;
  for (uint64_t base = s0; base < s1; base += stride) {
    const bool in = base + quad < s1;
    const uint64_t seed = listed ? (in ? list[base + quad] : 0) : base + quad;
    KEY key = in ? seed_key[seed] : key_invalid<KEY>();
    bool alive = key != key_invalid<KEY>();
    uint32_t l = 0, r = fm.n, j0 = 0;
    if (fm.ftab != nullptr && k >= fm.ftab_len) {
      // the first ftab_len steps (the seed's last ftab_len bases) are one table lookup
      j0 = fm.ftab_len;
      if (alive) {
        uint2 iv = fm.ftab[(uint64_t)key & ((1ull << (2 * j0)) - 1ull)];
        l = iv.x; r = iv.y;
        alive = r > l;
      }
    }
    // LF steps, per quad, until the seed is exhausted -- or until its interval is small and the
    // rest of the seed short enough to be checked against the text itself (whole SA resident)
    uint32_t jq = j0;
    while (true) {
      bool step = alive && jq < k && !(can_verify && (r - l) <= VERIFY_ROWS && (k - jq) <= 16u);
      if (!__any(step)) break;
      if (step) {
        uint32_t c = (uint32_t)(key >> (2 * jq)) & 3u;
        uint32_t bl = l / BLOCK_SYMS, br = r / BLOCK_SYMS;
        uint4 vl = fm.blocks[(uint64_t)bl * 4 + ql];
        uint4 vr = vl;
        if (br != bl) vr = fm.blocks[(uint64_t)br * 4 + ql];
        uint32_t nl = fm.C[c] + quad_rank(fm, s_sup, vl, ql, c, l);
        uint32_t nr = fm.C[c] + quad_rank(fm, s_sup, vr, ql, c, r);
        l = nl; r = nr;
        alive = r > l;
        ++jq;
        n_steps += ql == 0;
      }
    }
    // verification: the quad's lanes take the interval's rows four at a time
    uint32_t cnt = alive ? r - l : 0u, aux = 0;
    if (__any(alive && jq < k)) {
      uint32_t rem = k - jq, mask = 0;
      if (alive && jq < k) {
        for (uint32_t t = ql; t < r - l; t += 4)
          if (text_matches(fm.text4, fm.sa[l + t], rem, key, k)) mask |= 1u << t;
      }
      mask = quad_sum(mask);                       // disjoint bits: sum == or
      if (alive && jq < k) { n_rows += ql == 0 ? r - l : 0u; cnt = (uint32_t)__popc(mask); aux = (rem << 8) | mask; }
    }
    // seeds above the gocc threshold are dropped here (index_iter.hpp:843-847)
    bool keep = cnt != 0 && cnt <= gocc_thr;
    if (in && ql == 0) {
      iv_lo[seed] = l;
      iv_cnt[seed] = keep ? cnt : 0u;
      iv_aux[seed] = aux;
      n_live += keep;
      wsum += keep ? cnt : 0u;
      if (listed && keep) atomicAdd((unsigned long long*)&wave_total[seed / per_wave], (unsigned long long)cnt);
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_live += __shfl_down(n_live, d); wsum += __shfl_down(wsum, d);
    n_steps += __shfl_down(n_steps, d); n_rows += __shfl_down(n_rows, d);
  }
  if (lane_id() == 0) {
    if (!listed) wave_total[wave] = wsum;
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
    if (n_steps) ctr->n_lf_steps.add((unsigned long long)n_steps);
    if (n_rows) ctr->n_rows_verified.add((unsigned long long)n_rows);
  }
}

// exclusive scan of the per-wave totals (at most WAVES_MAX values): one workgroup of 1024 threads,
// WAVES_MAX / 1024 values per thread, wave shuffles + one LDS hop
constexpr int WAVES_MAX = 8192;       // waves of K1 / the probe / K2: all resident at once (16 K and 32 K measured slower)

__global__ void __launch_bounds__(1024)
k_wave_offsets(uint64_t* wave_total, const uint64_t* __restrict__ wave_total_off, uint64_t n_waves, uint64_t* total_on,
               uint64_t* total_all, bool accumulate)
{
  // accumulate: this is a further part of the index -- its hits go behind those already counted in *total_all.
  // wave_total[n_waves] / [n_waves + 1] get the part's range of output slots.
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t osum[16];
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  constexpr int V = WAVES_MAX / 1024;
  const uint64_t base = accumulate ? *total_all : 0, base_on = accumulate ? *total_on : 0;
  uint64_t v[V], s = 0, on = 0;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    uint64_t idx = (uint64_t)t * V + i;
    uint64_t a = idx < n_waves ? wave_total[idx] : 0;
    on += a;
    v[i] = a + ((wave_total_off && idx < n_waves) ? wave_total_off[idx] : 0);
    s += v[i];
  }
  for (int d = 32; d > 0; d >>= 1) on += __shfl_down(on, d);
  if (lane == 0) osum[w] = on;
  uint64_t incl = s;
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t u = __shfl_up(incl, d);
    if (lane >= (uint32_t)d) incl += u;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint64_t before = 0, all = 0;
  for (uint32_t i = 0; i < 16; ++i) { if (i < w) before += wsum[i]; all += wsum[i]; }
  uint64_t run = base + before + incl - s;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    uint64_t idx = (uint64_t)t * V + i;
    if (idx < n_waves) wave_total[idx] = run;
    run += v[i];
  }
  if (t == 0) {
    uint64_t o = 0;
    for (uint32_t i = 0; i < 16; ++i) o += osum[i];
    *total_on = base_on + o;
    *total_all = base + all;
    wave_total[n_waves] = base;
    wave_total[n_waves + 1] = base + all;
  }
}

// An index in several parts: K1 ran once per part and left every part's occurrence count per seed.  A gocc
// threshold counts a k-mer's occurrences in the whole path text (index_iter.hpp:843-847): seeds whose counts add
// up to more than `thr` lose them in every part; the per-wave totals of every part are made here (K1's own were
// taken before the threshold), and the seeds with an occurrence in any part are counted once.
__global__ void __launch_bounds__(256)
k_parts_combine(uint32_t* __restrict__ iv_cnt, uint64_t seed_stride, uint32_t n_parts, uint32_t thr,
                const uint64_t* __restrict__ params, uint64_t seeds_cap, uint32_t per_wave,
                uint64_t* __restrict__ wave_total, uint64_t tiles_stride, DevCounters* ctr)
{
  const uint32_t lane = lane_id();
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_seeds = min(params[0], seeds_cap);
  const uint64_t s0 = wave * per_wave, s1 = min(n_seeds, s0 + per_wave);
  uint64_t wsum[PSIGPU_MAX_PARTS];
  uint32_t n_live = 0;
  for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] = 0;
  for (uint64_t seed = s0 + lane; seed < s1; seed += 64) {
    uint64_t total = 0;
    uint32_t c[PSIGPU_MAX_PARTS];
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) { c[p] = p < n_parts ? iv_cnt[p * seed_stride + seed] : 0u; total += c[p]; }
    const bool keep = total != 0 && total <= thr;
    if (total != 0 && !keep)
      for (uint32_t p = 0; p < n_parts; ++p) if (c[p]) iv_cnt[p * seed_stride + seed] = 0;
    n_live += keep;
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] += keep ? c[p] : 0u;
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_live += __shfl_down(n_live, d);
#pragma unroll
    for (uint32_t p = 0; p < PSIGPU_MAX_PARTS; ++p) wsum[p] += __shfl_down(wsum[p], d);
  }
  if (lane == 0) {
    for (uint32_t p = 0; p < n_parts; ++p) wave_total[p * tiles_stride + wave] = wsum[p];
    if (n_live) ctr->n_live.add((unsigned long long)n_live);
  }
}

