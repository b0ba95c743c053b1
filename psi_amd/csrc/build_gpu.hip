// Device-side construction of the FM-index arrays from the indexed text:
// suffix array by prefix doubling, BWT rank blocks, exception list, SA samples, interval table
// and the 4-bit text -- bit-identical to what the host builder (index.cpp, SA-IS) produces,
// which is how it is tested (tests/test_gpu_build.py).
//
// The reference builds these through sdsl::construct (reference include/psi/fmindex.hpp:257-271).
// Here: suffixes are first sorted by their leading 10 symbols, then by (rank, rank of the suffix
// h further) with h doubling until every rank is unique (Manber-Myers / Larsson-Sadakane
// doubling, all suffixes every round).  The sort itself is rocPRIM's device radix sort -- the
// one library building block of this file; every other kernel is written here.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "host.hpp"
#include "loci_steps.hpp"

namespace psigpu {
namespace {

#define GB_CHK(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) { *err = std::string(#call) + ": " + hipGetErrorString(e_); return PSIGPU_ERR_DEVICE; } \
  } while (0)

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  void drop() { if (p) (void)hipFree(p); p = nullptr; }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

constexpr uint32_t W0 = 10;                 // symbols in the initial key (3 bits each)

__global__ void k_init(const uint8_t* __restrict__ T, uint32_t n, uint32_t* __restrict__ key, uint32_t* __restrict__ sa)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k = 0;
  for (uint32_t j = 0; j < W0; ++j) k = (k << 3) | (i + j < n ? T[i + j] : 0u);
  key[i] = k;
  sa[i] = i;
}

template <typename K>
__global__ void k_heads(const K* __restrict__ key, uint32_t n, uint32_t* __restrict__ head, uint32_t* __restrict__ flag)
{
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  bool f = j == 0 || key[j] != key[j - 1];
  head[j] = f ? j : 0u;
  flag[j] = f ? 1u : 0u;
}

__global__ void k_set_rank(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ head, uint32_t n,
                           uint32_t* __restrict__ rank)
{
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) rank[sa[j]] = head[j];
}

__global__ void k_key64(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ rank, uint32_t n, uint32_t h,
                        uint64_t* __restrict__ key)
{
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t s = sa[j];
  uint64_t second = (uint64_t)s + h < n ? (uint64_t)rank[s + h] + 1 : 0;
  key[j] = ((uint64_t)rank[s] << 32) | second;
}

__device__ __forceinline__ uint32_t bwt_sym(const uint8_t* T, const uint32_t* sa, uint32_t n, uint32_t i)
{
  uint32_t s = sa[i];
  return s ? T[s - 1] : T[n - 1];
}

// per 192-row block: number of A, C, G rows and of exception rows (separator / sentinel in the BWT)
__global__ void k_block_counts(const uint8_t* __restrict__ T, const uint32_t* __restrict__ sa, uint32_t n, uint32_t nblk,
                               uint32_t* __restrict__ cA, uint32_t* __restrict__ cC, uint32_t* __restrict__ cG,
                               uint32_t* __restrict__ cE, uint32_t* __restrict__ cT)
{
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  uint32_t a = 0, c = 0, g = 0, e = 0, t = 0;
  uint32_t lo = b * BLOCK_SYMS, hi = min(n, lo + BLOCK_SYMS);
  for (uint32_t i = lo; i < hi; ++i) {
    uint32_t s = bwt_sym(T, sa, n, i);
    a += s == SYM_A; c += s == SYM_C; g += s == SYM_G; t += s == SYM_T; e += s < SYM_A;
  }
  cA[b] = a; cC[b] = c; cG[b] = g; cE[b] = e; cT[b] = t;
}

__global__ void k_block_build(const uint8_t* __restrict__ T, const uint32_t* __restrict__ sa, uint32_t n, uint32_t nblk,
                              const uint32_t* __restrict__ pA, const uint32_t* __restrict__ pC,
                              const uint32_t* __restrict__ pG, const uint32_t* __restrict__ pE,
                              const uint32_t* __restrict__ cE, uint32_t exc_shift, RankBlock* __restrict__ blocks,
                              uint32_t* __restrict__ exc_row, uint32_t* __restrict__ exc_sa, uint32_t* __restrict__ exc_super)
{
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  RankBlock B;
  B.cnt[0] = pA[b]; B.cnt[1] = pC[b]; B.cnt[2] = pG[b];
  uint32_t ne = cE[b];
  // exceptions in front of the block, counted from its super-block's start (the header field has 24 bits)
  const uint32_t sb = (b >> exc_shift) << exc_shift;
  if (b == sb) exc_super[b >> exc_shift] = pE[b];
  B.exc = ((pE[b] - pE[sb]) << 8) | min(ne, 255u);
  for (int w = 0; w < 6; ++w) B.sym[w] = 0;
  uint32_t lo = b * BLOCK_SYMS, hi = min(n, lo + BLOCK_SYMS), e = pE[b];
  for (uint32_t i = lo; i < hi; ++i) {
    uint32_t s = bwt_sym(T, sa, n, i), j = i - lo;
    uint64_t two = 0;
    if (s >= SYM_A) two = s - SYM_A;
    else { exc_row[e] = i; exc_sa[e] = sa[i]; ++e; }
    B.sym[2 * (j >> 6)] |= (two & 1) << (j & 63);
    B.sym[2 * (j >> 6) + 1] |= (two >> 1) << (j & 63);
  }
  blocks[b] = B;
}

__global__ void k_samples(const uint32_t* __restrict__ sa, uint32_t n, uint32_t rate, uint32_t* __restrict__ out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((uint64_t)i * rate < n) out[i] = sa[i * rate];
}

// 2-bit code of T[p, p+q) (first base most significant), QNONE when a non-base symbol is inside
// (64-bit: at q = 16 every 32-bit value is the code of some q-mer)
constexpr uint64_t QNONE = ~0ull;
__device__ __forceinline__ uint64_t qcode(const uint8_t* T, uint32_t n, uint32_t p, uint32_t q)
{
  if ((uint64_t)p + q > n) return QNONE;
  uint64_t c = 0;
  for (uint32_t j = 0; j < q; ++j) {
    uint32_t s = T[p + j];
    if (s < SYM_A) return QNONE;
    c = (c << 2) | (s - SYM_A);
  }
  return c;
}

__global__ void k_ftab(const uint8_t* __restrict__ T, const uint32_t* __restrict__ sa, uint32_t n, uint32_t q,
                       uint2* __restrict__ ftab)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t c = qcode(T, n, sa[i], q);
  uint64_t cp = i ? qcode(T, n, sa[i - 1], q) : QNONE;
  if (c != cp) {
    if (c != QNONE) ftab[c].x = i;
    if (cp != QNONE) ftab[cp].y = i;
  }
  if (i == n - 1 && c != QNONE) ftab[c].y = n;
}

__global__ void k_text4(const uint8_t* __restrict__ T, uint32_t n, uint64_t nwords, uint64_t* __restrict__ out)
{
  uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  uint64_t v = 0;
  for (uint32_t j = 0; j < 16; ++j) {
    uint64_t i = w * 16 + j;
    if (i < n) {
      uint64_t nib = T[i] >= SYM_A ? (uint64_t)(T[i] - SYM_A) : 4ull;
      v |= nib << (60 - 4 * j);
    }
  }
  out[w] = v;
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

template <typename K>
int sort_pairs(K* kin, K* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned end_bit, Buf& tmp, size_t& tmp_cap,
               std::string* err)
{
  size_t need = 0;
  GB_CHK(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, n, 0, end_bit, 0));
  if (need > tmp_cap) {
    if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; }
    GB_CHK(tmp.alloc(need));
    tmp_cap = need;
  }
  GB_CHK(rocprim::radix_sort_pairs(tmp.p, need, kin, kout, vin, vout, n, 0, end_bit, 0));
  return PSIGPU_OK;
}

int scan_u32(uint32_t* in, uint32_t* out, size_t n, bool exclusive_sum, Buf& tmp, size_t& tmp_cap, std::string* err)
{
  size_t need = 0;
  if (exclusive_sum) {
    GB_CHK(rocprim::exclusive_scan(nullptr, need, in, out, 0u, n, rocprim::plus<uint32_t>(), 0));
  } else {
    GB_CHK(rocprim::inclusive_scan(nullptr, need, in, out, n, rocprim::maximum<uint32_t>(), 0));
  }
  if (need > tmp_cap) {
    if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; }
    GB_CHK(tmp.alloc(need));
    tmp_cap = need;
  }
  if (exclusive_sum) {
    GB_CHK(rocprim::exclusive_scan(tmp.p, need, in, out, 0u, n, rocprim::plus<uint32_t>(), 0));
  } else {
    GB_CHK(rocprim::inclusive_scan(tmp.p, need, in, out, n, rocprim::maximum<uint32_t>(), 0));
  }
  return PSIGPU_OK;
}

}  // namespace

// T: the text in builder coding (0 sentinel, 1 separator, 2..5 ACGT), sentinel last.
// Fills x->blocks, samples, exc_row, exc_sa, C, ftab, text4 (and *sa_out when requested).
// A large array from the device into ordinary (pageable) host memory: hipMemcpy moves such a copy through one staging
// thread (a whole-genome part, 33 GB of suffix array, interval table, rank blocks and text, took several seconds);
// here 32-MiB pieces land in two pinned buffers and a few threads copy each into place while the next one is on
// its way.  Small arrays and any failure to get pinned memory: the plain copy.
static hipError_t download(void* dst, const void* src, size_t bytes)
{
  constexpr size_t PIECE = 32u << 20;
  if (bytes < 2 * PIECE) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  struct Stage {
    void* buf[2] = { nullptr, nullptr };
    hipEvent_t done[2] = { nullptr, nullptr };
    hipStream_t s = nullptr;
    ~Stage()
    {
      if (s) (void)hipStreamSynchronize(s);       // (an error path may leave a piece in flight)
      for (int i = 0; i < 2; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (done[i]) (void)hipEventDestroy(done[i]); }
      if (s) (void)hipStreamDestroy(s);
    }
  } st;
  bool ok = hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipHostMalloc(&st.buf[i], PIECE, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&st.done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); }
  hipError_t e = hipDeviceSynchronize();          // (the producers ran on the default stream)
  if (e != hipSuccess) return e;
  auto copy_out = [&](size_t piece) {             // pinned piece -> its place, on a few threads
    const size_t off = piece * PIECE, len = std::min(PIECE, bytes - off);
    const char* from = (const char*)st.buf[piece & 1];
    char* to = (char*)dst + off;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned parts = std::min<unsigned>(8, hw ? hw : 1);
    std::vector<std::thread> th;
    const size_t per = (len / parts + 63) & ~(size_t)63;
    for (unsigned t = 1; t < parts; ++t) {
      const size_t a = std::min(len, t * per), b = std::min(len, (t + 1) * per);
      if (b > a) th.emplace_back([=] { memcpy(to + a, from + a, b - a); });
    }
    memcpy(to, from, std::min(len, per));
    for (auto& t : th) t.join();
  };
  const size_t n_pieces = (bytes + PIECE - 1) / PIECE;
  for (size_t piece = 0; piece <= n_pieces; ++piece) {
    if (piece < n_pieces) {                       // start piece `piece` (its buffer was emptied two iterations ago)
      const size_t off = piece * PIECE, len = std::min(PIECE, bytes - off);
      e = hipMemcpyAsync(st.buf[piece & 1], (const char*)src + off, len, hipMemcpyDeviceToHost, st.s);
      if (e == hipSuccess) e = hipEventRecord(st.done[piece & 1], st.s);
      if (e != hipSuccess) return e;
    }
    if (piece >= 1) {                             // while it is on its way, piece - 1 goes into place
      e = hipEventSynchronize(st.done[(piece - 1) & 1]);
      if (e != hipSuccess) return e;
      copy_out(piece - 1);
    }
  }
  return hipSuccess;
}

int gpu_build_fm(const std::vector<uint8_t>& T, uint32_t sa_rate, uint32_t q, int device, Index* x,
                 std::vector<int32_t>* sa_out, std::string* err)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    *err = "no such HIP device for the index build";
    return PSIGPU_ERR_DEVICE;
  }
  GB_CHK(hipSetDevice(device));
  const uint64_t n64 = T.size();
  if (n64 >= 0xFFFFFF00ull) { *err = "text too long for the 32-bit index layout"; return PSIGPU_ERR_ARG; }
  const uint32_t n = (uint32_t)n64;
  int st;
  Buf dT, sa0, sa1, k32a, k32b, k64a, k64b, rank, head, flag, tmp;
  size_t tmp_cap = 0;
  GB_CHK(dT.alloc(n + 64));
  GB_CHK(hipMemcpy(dT.p, T.data(), n, hipMemcpyHostToDevice));
  GB_CHK(sa0.alloc((size_t)n * 4)); GB_CHK(sa1.alloc((size_t)n * 4));
  GB_CHK(k32a.alloc((size_t)n * 4)); GB_CHK(k32b.alloc((size_t)n * 4));
  GB_CHK(rank.alloc((size_t)n * 4)); GB_CHK(head.alloc((size_t)n * 4)); GB_CHK(flag.alloc((size_t)n * 4));
  const unsigned g = grid_for(n);
  uint32_t* sa_cur = sa0.as<uint32_t>();
  uint32_t* sa_alt = sa1.as<uint32_t>();

  // ---- round 0: leading W0 symbols --------------------------------------------------------
  k_init<<<g, 256>>>(dT.as<uint8_t>(), n, k32a.as<uint32_t>(), sa_cur);
  if ((st = sort_pairs(k32a.as<uint32_t>(), k32b.as<uint32_t>(), sa_cur, sa_alt, n, 3 * W0, tmp, tmp_cap, err))) return st;
  std::swap(sa_cur, sa_alt);
  k_heads<uint32_t><<<g, 256>>>(k32b.as<uint32_t>(), n, head.as<uint32_t>(), flag.as<uint32_t>());
  auto finish_round = [&](uint64_t* groups) -> int {
    // group-start rank of every suffix, number of distinct groups
    int s2;
    if ((s2 = scan_u32(head.as<uint32_t>(), k32b.as<uint32_t>(), n, false, tmp, tmp_cap, err))) return s2;
    k_set_rank<<<g, 256>>>(sa_cur, k32b.as<uint32_t>(), n, rank.as<uint32_t>());
    size_t need = 0;
    uint32_t* d_sum = k32a.as<uint32_t>();       // scratch word
    GB_CHK(rocprim::reduce(nullptr, need, flag.as<uint32_t>(), d_sum, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0));
    if (need > tmp_cap) { if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; } GB_CHK(tmp.alloc(need)); tmp_cap = need; }
    GB_CHK(rocprim::reduce(tmp.p, need, flag.as<uint32_t>(), d_sum, 0u, (size_t)n, rocprim::plus<uint32_t>(), 0));
    uint32_t hsum = 0;
    GB_CHK(hipMemcpy(&hsum, d_sum, 4, hipMemcpyDeviceToHost));
    *groups = hsum;
    return PSIGPU_OK;
  };
  uint64_t groups = 0;
  if ((st = finish_round(&groups))) return st;

  // ---- doubling -----------------------------------------------------------------------------
  if (groups < n) {
    GB_CHK(k64a.alloc((size_t)n * 8)); GB_CHK(k64b.alloc((size_t)n * 8));
  }
  for (uint64_t h = W0; groups < n; h *= 2) {
    if (h > 2ull * n) { *err = "suffix sorting did not converge"; return PSIGPU_ERR_ARG; }
    k_key64<<<g, 256>>>(sa_cur, rank.as<uint32_t>(), n, (uint32_t)std::min<uint64_t>(h, n), k64a.as<uint64_t>());
    if ((st = sort_pairs(k64a.as<uint64_t>(), k64b.as<uint64_t>(), sa_cur, sa_alt, n, 64, tmp, tmp_cap, err))) return st;
    std::swap(sa_cur, sa_alt);
    k_heads<uint64_t><<<g, 256>>>(k64b.as<uint64_t>(), n, head.as<uint32_t>(), flag.as<uint32_t>());
    if ((st = finish_round(&groups))) return st;
  }
  GB_CHK(hipDeviceSynchronize());
  if (sa_out && n < 0x7FFFFFF0u) {
    resize_populated(*sa_out, n);
    GB_CHK(download(sa_out->data(), sa_cur, (size_t)n * 4));
  }

  {
  // ---- BWT rank blocks + exceptions ------------------------------------------------------------
  const uint32_t nblk = n / BLOCK_SYMS + 1;
  Buf cA, cC, cG, cE, cT, pA, pC, pG, pE, dblocks, dexc_row, dexc_sa;
  for (Buf* b : { &cA, &cC, &cG, &cE, &cT, &pA, &pC, &pG, &pE }) GB_CHK(b->alloc((size_t)(nblk + 1) * 4));
  k_block_counts<<<grid_for(nblk), 256>>>(dT.as<uint8_t>(), sa_cur, n, nblk, cA.as<uint32_t>(), cC.as<uint32_t>(),
                                          cG.as<uint32_t>(), cE.as<uint32_t>(), cT.as<uint32_t>());
  if ((st = scan_u32(cA.as<uint32_t>(), pA.as<uint32_t>(), nblk, true, tmp, tmp_cap, err))) return st;
  if ((st = scan_u32(cC.as<uint32_t>(), pC.as<uint32_t>(), nblk, true, tmp, tmp_cap, err))) return st;
  if ((st = scan_u32(cG.as<uint32_t>(), pG.as<uint32_t>(), nblk, true, tmp, tmp_cap, err))) return st;
  if ((st = scan_u32(cE.as<uint32_t>(), pE.as<uint32_t>(), nblk, true, tmp, tmp_cap, err))) return st;
  // totals: prefix of the last block + its own count
  uint32_t lastp[4], lastc[5];
  Buf* ps[4] = { &pA, &pC, &pG, &pE };
  Buf* cs[5] = { &cA, &cC, &cG, &cE, &cT };
  for (int i = 0; i < 4; ++i) GB_CHK(hipMemcpy(&lastp[i], ps[i]->as<uint32_t>() + (nblk - 1), 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < 5; ++i) GB_CHK(hipMemcpy(&lastc[i], cs[i]->as<uint32_t>() + (nblk - 1), 4, hipMemcpyDeviceToHost));
  const uint64_t totA = (uint64_t)lastp[0] + lastc[0], totC = (uint64_t)lastp[1] + lastc[1];
  const uint64_t totG = (uint64_t)lastp[2] + lastc[2], totE = (uint64_t)lastp[3] + lastc[3];
  const uint32_t xs = x->exc_shift;
  const size_t n_super = ((size_t)(nblk - 1) >> xs) + 1;
  Buf dsuper;
  GB_CHK(dblocks.alloc((size_t)nblk * sizeof(RankBlock)));
  GB_CHK(dexc_row.alloc((size_t)(totE + 1) * 4)); GB_CHK(dexc_sa.alloc((size_t)(totE + 1) * 4));
  GB_CHK(dsuper.alloc(n_super * 4));
  k_block_build<<<grid_for(nblk), 256>>>(dT.as<uint8_t>(), sa_cur, n, nblk, pA.as<uint32_t>(), pC.as<uint32_t>(),
                                         pG.as<uint32_t>(), pE.as<uint32_t>(), cE.as<uint32_t>(), xs,
                                         dblocks.as<RankBlock>(), dexc_row.as<uint32_t>(), dexc_sa.as<uint32_t>(),
                                         dsuper.as<uint32_t>());
  x->exc_super.resize(n_super);
  GB_CHK(hipMemcpy(x->exc_super.data(), dsuper.p, n_super * 4, hipMemcpyDeviceToHost));
  resize_populated(x->blocks, nblk);
  GB_CHK(download(x->blocks.data(), dblocks.p, (size_t)nblk * sizeof(RankBlock)));
  x->exc_row.resize(totE); x->exc_sa.resize(totE);
  if (totE) {
    GB_CHK(hipMemcpy(x->exc_row.data(), dexc_row.p, totE * 4, hipMemcpyDeviceToHost));
    GB_CHK(hipMemcpy(x->exc_sa.data(), dexc_sa.p, totE * 4, hipMemcpyDeviceToHost));
  }
  // C[c] = symbols smaller than c: the sentinel and the separators are exactly the exceptions
  x->C[0] = totE;
  x->C[1] = x->C[0] + totA;
  x->C[2] = x->C[1] + totC;
  x->C[3] = x->C[2] + totG;

  }
  // ---- SA samples ---------------------------------------------------------------------------------
  const uint64_t nsamp = ((uint64_t)n + sa_rate - 1) / sa_rate;
  resize_populated(x->samples, nsamp);
  if (sa_rate == 1) {
    GB_CHK(download(x->samples.data(), sa_cur, (size_t)n * 4));
  } else {
    Buf ds;
    GB_CHK(ds.alloc(nsamp * 4));
    k_samples<<<grid_for(nsamp), 256>>>(sa_cur, n, sa_rate, ds.as<uint32_t>());
    GB_CHK(download(x->samples.data(), ds.p, nsamp * 4));
  }

  // ---- interval table --------------------------------------------------------------------------------
  x->ftab_len = q;
  x->ftab.clear();
  if (q) {
    const uint64_t entries = 1ull << (2 * q);
    Buf df;
    GB_CHK(df.alloc(entries * 8));
    GB_CHK(hipMemset(df.p, 0, entries * 8));
    k_ftab<<<g, 256>>>(dT.as<uint8_t>(), sa_cur, n, q, df.as<uint2>());
    resize_populated(x->ftab, 2 * entries);
    GB_CHK(download(x->ftab.data(), df.p, entries * 8));
  }

  // ---- 4-bit text ------------------------------------------------------------------------------------------
  const uint64_t nwords = (uint64_t)n / 16 + 2;
  {
    Buf dt4;
    GB_CHK(dt4.alloc(nwords * 8));
    k_text4<<<grid_for(nwords), 256>>>(dT.as<uint8_t>(), n, nwords, dt4.as<uint64_t>());
    resize_populated(x->text4, nwords);
    GB_CHK(download(x->text4.data(), dt4.p, nwords * 8));
  }
  GB_CHK(hipDeviceSynchronize());
  return PSIGPU_OK;
}

// (k-mer, locus) pairs of the locus k-mer table (device.hip): sort by the low `end_bit` key bits.
// Device pointers; runs on the null stream and returns when the sort is done.
int gpu_sort_pairs_u64(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, uint64_t n,
                       unsigned end_bit, std::string* err)
{
  Buf tmp;
  size_t tmp_cap = 0;
  int st = sort_pairs<uint64_t>(keys_in, keys_out, vals_in, vals_out, (size_t)n, end_bit, tmp, tmp_cap, err);
  if (st != PSIGPU_OK) return st;
  GB_CHK(hipDeviceSynchronize());
  return PSIGPU_OK;
}

// running maximum of a u64 array, in place (the path k-mers of the direct k-mer table build, device.hip)
int gpu_running_max_u64(uint64_t* data, uint64_t n, std::string* err)
{
  if (n == 0) return PSIGPU_OK;
  Buf tmp;
  size_t need = 0;
  GB_CHK(rocprim::inclusive_scan(nullptr, need, data, data, (size_t)n, rocprim::maximum<uint64_t>(), 0));
  GB_CHK(tmp.alloc(need));
  GB_CHK(rocprim::inclusive_scan(tmp.p, need, data, data, (size_t)n, rocprim::maximum<uint64_t>(), 0));
  GB_CHK(hipDeviceSynchronize());
  return PSIGPU_OK;
}

// ------------------------------------------------------------------------------------
// Starting loci on the device (reference add_uncovered_loci,
// include/psi/seed_finder.hpp:1481-1541, and add_all_loci :1543-1585 when no path is indexed).
// Same definition and the same result, in the same order, as find_starting_loci() in index.cpp
// (tests/test_gpu_build.py compares the two): a locus (v, o) is a starting locus iff some k-walk
// from it is not a contiguous run of an indexed path; coverage is one bit per (simple) path on
// nodes and edges.
// ------------------------------------------------------------------------------------
namespace {

__global__ void k_loci_len(const uint64_t* __restrict__ label_off, uint64_t n, uint32_t k, uint32_t* __restrict__ len,
                           uint32_t* __restrict__ reach)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  uint64_t l = label_off[v + 1] - label_off[v];
  len[v] = (uint32_t)min(l, (uint64_t)0xFFFFFFFFu);
  reach[v] = (uint32_t)min(l, (uint64_t)k);
}

// one relaxation sweep of reach(u) = min(k, len(u) + max_child reach(child)); child[] = that max
__global__ void k_loci_reach(const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ edge_to,
                             const uint32_t* __restrict__ len, uint64_t n, uint32_t k, const uint32_t* __restrict__ reach_in,
                             uint32_t* __restrict__ reach_out, uint32_t* __restrict__ child, uint32_t* __restrict__ changed)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  uint32_t best = 0;
  for (uint64_t e = edge_off[v]; e < edge_off[v + 1]; ++e) best = max(best, reach_in[edge_to[e]]);
  uint32_t r = (uint32_t)min((uint64_t)k, (uint64_t)len[v] + best);
  r = max(r, reach_in[v]);
  child[v] = best;
  reach_out[v] = r;
  if (r != reach_in[v]) *changed = 1;
}

// does path p visit a node twice?  (seen[] holds the last path that touched a node)
__global__ void k_loci_simple(const uint32_t* __restrict__ path, uint64_t len, uint32_t p, uint32_t* __restrict__ seen,
                              uint32_t* __restrict__ dup)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  if (atomicExch(&seen[path[i]], p) == p) *dup = 1;
}

__global__ void k_loci_masks(const uint32_t* __restrict__ path, uint64_t len, uint64_t bit,
                             const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ edge_to,
                             unsigned long long* __restrict__ node_mask, unsigned long long* __restrict__ edge_mask)
{
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  const uint32_t v = path[i];
  atomicOr(&node_mask[v], (unsigned long long)bit);
  if (i + 1 < len) {
    const uint32_t w = path[i + 1];
    for (uint64_t e = edge_off[v]; e < edge_off[v + 1]; ++e)
      if (edge_to[e] == w) { atomicOr(&edge_mask[e], (unsigned long long)bit); break; }
  }
}

struct LociGraph {
  const uint64_t* edge_off; const uint32_t* edge_to; const uint32_t* len;
  const uint32_t* reach; const uint32_t* child;
  const unsigned long long* node_mask; const unsigned long long* edge_mask;
  uint64_t n; uint32_t k, step;
};

constexpr int LOCI_DEPTH = 4 * 63 + 3;       // index.cpp's explore(): depth <= 4 k (k <= PSIGPU_MAX_SEED_LEN)

// bit `need` (1..k-1) of the result: an uncovered walk exists that leaves node v after `need`
// more bases are still wanted -- explore() of index.cpp with an explicit stack
__device__ uint64_t loci_uncovered(const LociGraph& g, uint64_t v)
{
  struct Frame { uint32_t e, e_end, S; unsigned long long mask; };
  Frame st[LOCI_DEPTH];
  int sp = 0;
  uint64_t unc = 0;
  st[sp++] = Frame{ (uint32_t)g.edge_off[v], (uint32_t)g.edge_off[v + 1], 0u, g.node_mask[v] };
  while (sp) {
    Frame& f = st[sp - 1];
    if (f.e == f.e_end) { --sp; continue; }
    const uint32_t e = f.e++;
    const uint32_t depth = (uint32_t)sp - 1;
    if (depth > 4 * g.k) continue;            // guards cycles of empty nodes
    const uint32_t u = g.edge_to[e];
    const unsigned long long m = f.mask & g.edge_mask[e] & g.node_mask[u];
    if (m == 0) {
      // entering u makes the walk uncovered: any need in [S+1, S+reach(u)] completes inside/after u
      uint32_t lo = f.S + 1, hi = min(f.S + g.reach[u], g.k - 1);
      if (hi >= lo) unc |= ((hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull));
      continue;
    }
    const uint32_t S2 = f.S + g.len[u];
    if (S2 >= g.k - 1) continue;
    if (sp < LOCI_DEPTH) { const uint32_t S1 = S2; st[sp++] = Frame{ (uint32_t)g.edge_off[u], (uint32_t)g.edge_off[u + 1], S1, m }; }
  }
  return unc;
}

// loci of node v, in offset order: counted (out == nullptr) or written at out + first[v]
__device__ __forceinline__ uint32_t loci_of_node(const LociGraph& g, uint64_t v, uint64_t unc, uint32_t* out_n, uint32_t* out_o)
{
  const uint64_t len = g.len[v], k = g.k, child = g.child[v];
  if (len == 0 || len + child < k) return 0;
  const bool node_unc = g.node_mask[v] == 0;
  const uint64_t o_max = len + child - k;                      // last offset with a k-walk
  uint64_t o = 0;
  if (!node_unc) o = len >= k ? len - k + 1 : 0;               // need = k - (len - o) > 0
  uint32_t since = 0, cnt = 0;
  for (; o <= o_max && o < len; ++o) {
    bool take = node_unc;
    if (!take) { const uint64_t need = k - (len - o); take = (unc >> need) & 1; }
    if (!take) continue;
    if (since % g.step == 0) {
      if (out_n) { out_n[cnt] = (uint32_t)v; out_o[cnt] = (uint32_t)o; }
      ++cnt;
    }
    ++since;
  }
  return cnt;
}

__global__ void k_loci_count(LociGraph g, unsigned long long* __restrict__ unc_out, uint32_t* __restrict__ cnt)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.n) return;
  unsigned long long unc = 0;
  if (g.len[v] != 0 && g.node_mask[v] != 0) unc = loci_uncovered(g, v);
  unc_out[v] = unc;
  cnt[v] = loci_of_node(g, v, unc, nullptr, nullptr);
}

__global__ void k_loci_fill(LociGraph g, const unsigned long long* __restrict__ unc, const uint32_t* __restrict__ first,
                            const uint32_t* __restrict__ cnt, uint32_t* __restrict__ out_n, uint32_t* __restrict__ out_o)
{
  uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.n || cnt[v] == 0) return;
  loci_of_node(g, v, unc[v], out_n + first[v], out_o + first[v]);
}

// ------------------------------------------------------------------------------------
// Starting loci of TRIMMED paths (psikt's default: patches), of more than 64 paths, of paths that visit a node
// twice -- where one coverage bit per path does not do.  The host routine's own scheme (index.cpp
// find_starting_loci / explore): a walk is covered while it is a run of consecutive STEPS of some path, a step
// indexing bases [lo, hi) of its node; per node the candidate steps of a walk depend on its start offset only
// through the distinct head offsets of the steps at the node.  One thread per node, explore() with an explicit
// stack and the candidate lists in a per-thread pool.  A node whose lists do not fit the pool is counted in
// `hard`; the caller then takes the host routine (same result either way; the fuzz campaigns build with
// PSIGPU_BUILD_VERIFY=1, which compares the two).
// ------------------------------------------------------------------------------------
__global__ void k_steps_init(const uint32_t* __restrict__ step_node, const uint32_t* __restrict__ len, uint64_t total,
                             uint32_t* __restrict__ lo, uint32_t* __restrict__ hi, uint8_t* __restrict__ last, uint32_t* __restrict__ cnt)
{
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  const uint32_t v = step_node[s];
  lo[s] = 0; hi[s] = len[v]; last[s] = 0;
  atomicAdd(&cnt[v], 1u);
}
__global__ void k_steps_ends(const uint64_t* __restrict__ first, uint64_t n_paths, const uint32_t* __restrict__ head,
                             const uint32_t* __restrict__ tail, const uint32_t* __restrict__ step_node, const uint32_t* __restrict__ len,
                             uint32_t* __restrict__ lo, uint32_t* __restrict__ hi, uint8_t* __restrict__ last)
{
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_paths || first[p + 1] == first[p]) return;
  const uint64_t s0 = first[p], s1 = first[p + 1] - 1;
  if (head) lo[s0] = min(len[step_node[s0]], head[p]);
  if (tail && tail[p]) hi[s1] = min(len[step_node[s1]], tail[p]);
  last[s1] = 1;
}
__global__ void k_iota_u32(uint32_t* __restrict__ a, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (uint32_t)i;
}

__global__ void k_steps_loci_count(StepGraph g, uint32_t* __restrict__ cnt, uint32_t* __restrict__ n_hard)
{
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.n) return;
  bool hard = false;
  const uint32_t c = steps_loci_of_node(g, v, nullptr, nullptr, &hard);
  cnt[v] = hard ? 0u : c;
  if (hard) atomicAdd(n_hard, 1u);
}
__global__ void k_steps_loci_fill(StepGraph g, const uint32_t* __restrict__ first, const uint32_t* __restrict__ cnt,
                                  uint32_t* __restrict__ out_n, uint32_t* __restrict__ out_o)
{
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.n || cnt[v] == 0) return;
  bool hard = false;
  steps_loci_of_node(g, v, out_n + first[v], out_o + first[v], &hard);
}

}  // namespace

int gpu_find_starting_loci(const Graph& g, const std::vector<std::vector<uint32_t>>& paths, uint32_t k, uint32_t step,
                           int device, std::vector<uint32_t>& loci_node, std::vector<uint32_t>& loci_off, std::string* err)
{
  loci_node.clear();
  loci_off.clear();
  if (step == 0) step = 1;
  const uint64_t n = g.n_nodes(), m = g.edge_to.size();
  if (n == 0) return PSIGPU_OK;
  GB_CHK(hipSetDevice(device));
  Buf d_label_off, d_edge_off, d_edge_to, d_len, d_reach_a, d_reach_b, d_child, d_node_mask, d_edge_mask, d_flag, d_seen,
      d_path, d_unc, d_cnt, d_first, d_out_n, d_out_o, tmp;
  size_t tmp_cap = 0;
  GB_CHK(d_label_off.alloc((n + 1) * 8)); GB_CHK(d_edge_off.alloc((n + 1) * 8)); GB_CHK(d_edge_to.alloc((m + 1) * 4));
  GB_CHK(hipMemcpy(d_label_off.p, g.label_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  GB_CHK(hipMemcpy(d_edge_off.p, g.edge_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  if (m) GB_CHK(hipMemcpy(d_edge_to.p, g.edge_to.data(), m * 4, hipMemcpyHostToDevice));
  GB_CHK(d_len.alloc(n * 4)); GB_CHK(d_reach_a.alloc(n * 4)); GB_CHK(d_reach_b.alloc(n * 4)); GB_CHK(d_child.alloc(n * 4));
  GB_CHK(d_flag.alloc(16));
  const unsigned gn = grid_for(n);
  k_loci_len<<<gn, 256>>>(d_label_off.as<uint64_t>(), n, k, d_len.as<uint32_t>(), d_reach_a.as<uint32_t>());
  d_label_off.drop();
  // reach / child: relax until nothing changes (one more sweep than needed leaves child[] final)
  uint32_t* ra = d_reach_a.as<uint32_t>();
  uint32_t* rb = d_reach_b.as<uint32_t>();
  for (;;) {
    GB_CHK(hipMemset(d_flag.p, 0, 4));
    k_loci_reach<<<gn, 256>>>(d_edge_off.as<uint64_t>(), d_edge_to.as<uint32_t>(), d_len.as<uint32_t>(), n, k, ra, rb,
                              d_child.as<uint32_t>(), d_flag.as<uint32_t>());
    uint32_t changed = 0;
    GB_CHK(hipMemcpy(&changed, d_flag.p, 4, hipMemcpyDeviceToHost));
    std::swap(ra, rb);
    if (!changed) break;
  }
  // coverage bits of the first 64 simple paths
  GB_CHK(d_node_mask.alloc(n * 8)); GB_CHK(d_edge_mask.alloc((m + 1) * 8)); GB_CHK(d_seen.alloc(n * 4));
  GB_CHK(hipMemset(d_node_mask.p, 0, n * 8)); GB_CHK(hipMemset(d_edge_mask.p, 0, (m + 1) * 8));
  GB_CHK(hipMemset(d_seen.p, 0xFF, n * 4));
  size_t longest = 0;
  for (auto& P : paths) longest = std::max(longest, P.size());
  GB_CHK(d_path.alloc((longest + 1) * 4));
  uint32_t bit = 0;
  for (size_t p = 0; p < paths.size() && bit < 64; ++p) {
    const auto& P = paths[p];
    if (P.empty()) { ++bit; continue; }              // (an empty path is simple and takes a bit, as on the host)
    GB_CHK(hipMemcpy(d_path.p, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    GB_CHK(hipMemset(d_flag.p, 0, 4));
    k_loci_simple<<<grid_for(P.size()), 256>>>(d_path.as<uint32_t>(), P.size(), (uint32_t)p, d_seen.as<uint32_t>(),
                                               d_flag.as<uint32_t>());
    uint32_t dup = 0;
    GB_CHK(hipMemcpy(&dup, d_flag.p, 4, hipMemcpyDeviceToHost));
    if (dup) continue;
    k_loci_masks<<<grid_for(P.size()), 256>>>(d_path.as<uint32_t>(), P.size(), 1ull << bit, d_edge_off.as<uint64_t>(),
                                              d_edge_to.as<uint32_t>(), d_node_mask.as<unsigned long long>(),
                                              d_edge_mask.as<unsigned long long>());
    ++bit;
  }
  // per node: uncovered extension lengths, number of loci; scan; fill
  GB_CHK(d_unc.alloc(n * 8)); GB_CHK(d_cnt.alloc((n + 1) * 4)); GB_CHK(d_first.alloc((n + 1) * 4));
  LociGraph lg = { d_edge_off.as<uint64_t>(), d_edge_to.as<uint32_t>(), d_len.as<uint32_t>(), ra, d_child.as<uint32_t>(),
                   d_node_mask.as<unsigned long long>(), d_edge_mask.as<unsigned long long>(), n, k, step };
  k_loci_count<<<gn, 256>>>(lg, d_unc.as<unsigned long long>(), d_cnt.as<uint32_t>());
  GB_CHK(hipMemset(d_cnt.as<uint32_t>() + n, 0, 4));
  int st = scan_u32(d_cnt.as<uint32_t>(), d_first.as<uint32_t>(), n + 1, true, tmp, tmp_cap, err);
  if (st != PSIGPU_OK) return st;
  uint32_t total = 0;
  GB_CHK(hipMemcpy(&total, d_first.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
  // (a 32-bit scan: more than 2^32 loci cannot be indexed by the u32 locus ids used downstream either)
  resize_populated(loci_node, total);
  resize_populated(loci_off, total);
  if (total) {
    GB_CHK(d_out_n.alloc((size_t)total * 4)); GB_CHK(d_out_o.alloc((size_t)total * 4));
    k_loci_fill<<<gn, 256>>>(lg, d_unc.as<unsigned long long>(), d_first.as<uint32_t>(), d_cnt.as<uint32_t>(),
                             d_out_n.as<uint32_t>(), d_out_o.as<uint32_t>());
    GB_CHK(download(loci_node.data(), d_out_n.p, (size_t)total * 4));
    GB_CHK(download(loci_off.data(), d_out_o.p, (size_t)total * 4));
  }
  GB_CHK(hipDeviceSynchronize());
  return PSIGPU_OK;
}

// The same for any set of paths (trimmed, many, not simple): see steps_loci_of_node.  *n_hard_out != 0: some nodes were
// left to the host -- the caller runs find_starting_loci() instead and the arrays here are to be ignored.
int gpu_find_starting_loci_steps(const Graph& g, const std::vector<std::vector<uint32_t>>& paths,
                                 const std::vector<uint32_t>& path_head, const std::vector<uint32_t>& path_tail,
                                 uint32_t k, uint32_t step, int device, std::vector<uint32_t>& loci_node,
                                 std::vector<uint32_t>& loci_off, uint64_t* n_hard_out, std::string* err)
{
  loci_node.clear();
  loci_off.clear();
  *n_hard_out = 0;
  if (step == 0) step = 1;
  const uint64_t n = g.n_nodes(), m = g.edge_to.size();
  if (n == 0) return PSIGPU_OK;
  std::vector<uint64_t> first(paths.size() + 1, 0);
  for (size_t p = 0; p < paths.size(); ++p) first[p + 1] = first[p] + paths[p].size();
  const uint64_t total = first[paths.size()];
  if (total >= 0xFFFFFFF0ull || n >= 0xFFFFFFF0ull) { *n_hard_out = 1; return PSIGPU_OK; }      // (32-bit step numbers here: the host's job)
  GB_CHK(hipSetDevice(device));
  Buf d_label_off, d_edge_off, d_edge_to, d_len, d_reach_a, d_reach_b, d_child, d_flag, d_first, d_head, d_tail,
      d_step_node, d_lo, d_hi, d_last, d_cnt, d_at_off, d_idx, d_keys_out, d_at, d_lcnt, d_lfirst, d_out_n, d_out_o, tmp;
  size_t tmp_cap = 0;
  GB_CHK(d_label_off.alloc((n + 1) * 8)); GB_CHK(d_edge_off.alloc((n + 1) * 8)); GB_CHK(d_edge_to.alloc((m + 1) * 4));
  GB_CHK(hipMemcpy(d_label_off.p, g.label_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  GB_CHK(hipMemcpy(d_edge_off.p, g.edge_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  if (m) GB_CHK(hipMemcpy(d_edge_to.p, g.edge_to.data(), m * 4, hipMemcpyHostToDevice));
  GB_CHK(d_len.alloc(n * 4)); GB_CHK(d_reach_a.alloc(n * 4)); GB_CHK(d_reach_b.alloc(n * 4)); GB_CHK(d_child.alloc(n * 4));
  GB_CHK(d_flag.alloc(16));
  const unsigned gn = grid_for(n);
  k_loci_len<<<gn, 256>>>(d_label_off.as<uint64_t>(), n, k, d_len.as<uint32_t>(), d_reach_a.as<uint32_t>());
  d_label_off.drop();
  uint32_t* ra = d_reach_a.as<uint32_t>();
  uint32_t* rb = d_reach_b.as<uint32_t>();
  for (;;) {
    GB_CHK(hipMemset(d_flag.p, 0, 4));
    k_loci_reach<<<gn, 256>>>(d_edge_off.as<uint64_t>(), d_edge_to.as<uint32_t>(), d_len.as<uint32_t>(), n, k, ra, rb,
                              d_child.as<uint32_t>(), d_flag.as<uint32_t>());
    uint32_t changed = 0;
    GB_CHK(hipMemcpy(&changed, d_flag.p, 4, hipMemcpyDeviceToHost));
    std::swap(ra, rb);
    if (!changed) break;
  }
  d_reach_a.drop(); d_reach_b.drop();
  // the steps: the paths' node lists one after the other, the bases each step indexes, the steps at every node
  GB_CHK(d_step_node.alloc((total + 1) * 4)); GB_CHK(d_lo.alloc((total + 1) * 4)); GB_CHK(d_hi.alloc((total + 1) * 4));
  GB_CHK(d_last.alloc(total + 1)); GB_CHK(d_cnt.alloc((n + 1) * 4)); GB_CHK(d_at_off.alloc((n + 1) * 4));
  {
    // (one transfer: patches are millions of short paths)
    std::vector<uint32_t> cat;
    resize_populated(cat, total);
    for (size_t p = 0; p < paths.size(); ++p)
      if (!paths[p].empty()) memcpy(cat.data() + first[p], paths[p].data(), paths[p].size() * 4);
    if (total) GB_CHK(hipMemcpy(d_step_node.p, cat.data(), total * 4, hipMemcpyHostToDevice));
  }
  GB_CHK(hipMemset(d_cnt.p, 0, (n + 1) * 4));
  if (total) {
    k_steps_init<<<grid_for(total), 256>>>(d_step_node.as<uint32_t>(), d_len.as<uint32_t>(), total, d_lo.as<uint32_t>(),
                                           d_hi.as<uint32_t>(), d_last.as<uint8_t>(), d_cnt.as<uint32_t>());
    GB_CHK(d_first.alloc(first.size() * 8));
    GB_CHK(hipMemcpy(d_first.p, first.data(), first.size() * 8, hipMemcpyHostToDevice));
    const bool have_head = path_head.size() >= paths.size(), have_tail = path_tail.size() >= paths.size();
    if (have_head) { GB_CHK(d_head.alloc(paths.size() * 4)); GB_CHK(hipMemcpy(d_head.p, path_head.data(), paths.size() * 4, hipMemcpyHostToDevice)); }
    if (have_tail) { GB_CHK(d_tail.alloc(paths.size() * 4)); GB_CHK(hipMemcpy(d_tail.p, path_tail.data(), paths.size() * 4, hipMemcpyHostToDevice)); }
    k_steps_ends<<<grid_for(paths.size()), 256>>>(d_first.as<uint64_t>(), paths.size(), have_head ? d_head.as<uint32_t>() : nullptr,
                                                  have_tail ? d_tail.as<uint32_t>() : nullptr, d_step_node.as<uint32_t>(),
                                                  d_len.as<uint32_t>(), d_lo.as<uint32_t>(), d_hi.as<uint32_t>(), d_last.as<uint8_t>());
  }
  int st = scan_u32(d_cnt.as<uint32_t>(), d_at_off.as<uint32_t>(), n + 1, true, tmp, tmp_cap, err);
  if (st != PSIGPU_OK) return st;
  GB_CHK(d_at.alloc((total + 1) * 4));
  if (total) {
    // a stable sort of the step numbers by node: a node's steps come out ascending
    GB_CHK(d_idx.alloc(total * 4)); GB_CHK(d_keys_out.alloc(total * 4));
    k_iota_u32<<<grid_for(total), 256>>>(d_idx.as<uint32_t>(), total);
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) < n) ++bits;
    st = sort_pairs<uint32_t>(d_step_node.as<uint32_t>(), d_keys_out.as<uint32_t>(), d_idx.as<uint32_t>(), d_at.as<uint32_t>(), total, bits,
                              tmp, tmp_cap, err);
    if (st != PSIGPU_OK) return st;
    d_idx.drop(); d_keys_out.drop();
  }
  StepGraph sg = { d_edge_off.as<uint64_t>(), d_edge_to.as<uint32_t>(), d_len.as<uint32_t>(), d_child.as<uint32_t>(),
                   d_step_node.as<uint32_t>(), d_lo.as<uint32_t>(), d_hi.as<uint32_t>(), d_last.as<uint8_t>(),
                   d_at_off.as<uint32_t>(), d_at.as<uint32_t>(), n, k, step };
  GB_CHK(d_lcnt.alloc((n + 1) * 4)); GB_CHK(d_lfirst.alloc((n + 1) * 4));
  GB_CHK(hipMemset(d_flag.p, 0, 4));
  k_steps_loci_count<<<grid_for(n), 256>>>(sg, d_lcnt.as<uint32_t>(), d_flag.as<uint32_t>());
  uint32_t n_hard = 0;
  GB_CHK(hipMemcpy(&n_hard, d_flag.p, 4, hipMemcpyDeviceToHost));
  if (n_hard) { *n_hard_out = n_hard; return PSIGPU_OK; }
  GB_CHK(hipMemset(d_lcnt.as<uint32_t>() + n, 0, 4));
  st = scan_u32(d_lcnt.as<uint32_t>(), d_lfirst.as<uint32_t>(), n + 1, true, tmp, tmp_cap, err);
  if (st != PSIGPU_OK) return st;
  uint32_t total_loci = 0;
  GB_CHK(hipMemcpy(&total_loci, d_lfirst.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
  resize_populated(loci_node, total_loci);
  resize_populated(loci_off, total_loci);
  if (total_loci) {
    GB_CHK(d_out_n.alloc((size_t)total_loci * 4)); GB_CHK(d_out_o.alloc((size_t)total_loci * 4));
    k_steps_loci_fill<<<grid_for(n), 256>>>(sg, d_lfirst.as<uint32_t>(), d_lcnt.as<uint32_t>(), d_out_n.as<uint32_t>(),
                                           d_out_o.as<uint32_t>());
    GB_CHK(download(loci_node.data(), d_out_n.p, (size_t)total_loci * 4));
    GB_CHK(download(loci_off.data(), d_out_o.p, (size_t)total_loci * 4));
  }
  GB_CHK(hipDeviceSynchronize());
  return PSIGPU_OK;
}

}  // namespace psigpu

