// Sort-unique of hit records on the device, in the order psikt writes them:
// (read_id, read_offset, node_id, node_offset).
//
// The reference emits an unordered multiset with duplicates (same locus found on several indexed
// paths, found both on a path and by the traverser, several walks spelling one k-mer:
// SURVEY 8a, include/psi/index_iter.hpp:662-677,728-746); parity is on the sort-unique set, and
// that is what PSIGPU_SORT_UNIQUE returns.  A hit is four u64 but carries far fewer bits: the read
// index inside the chunk, an offset inside a read, a node (by the rank of its id) and an offset
// inside a node.  Each record is packed into ONE 64-bit key with exactly those bits, most
// significant field first; the keys are radix-sorted over the bits in use only (rocPRIM, the one
// library building block), adjacent duplicates dropped, and the survivors expanded back into
// 32-byte records.  Everything is a coalesced stream over 8-byte keys: HBM-bandwidth work.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdint>
#include <string>

#include "host.hpp"

namespace psigpu {
namespace {

struct KeyLayout {
  uint32_t noff_bits, node_bits, roff_bits, rid_bits;
  uint64_t rec_offset;          // read ids are stored relative to the chunk
  uint64_t id_base;             // affine ids: id = id_base + rank
  const uint64_t* ids_sorted;   // otherwise: the node ids in increasing order (key holds the index)
  uint64_t n_nodes;
};

__device__ __forceinline__ uint64_t id_to_order(const KeyLayout& L, uint64_t id)
{
  if (L.ids_sorted == nullptr) return id - L.id_base;
  uint64_t lo = 0, hi = L.n_nodes;            // first index with ids_sorted[i] >= id
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (L.ids_sorted[mid] < id) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
k_hits_pack(const psigpu_hit* __restrict__ hits, uint64_t n, KeyLayout L, uint64_t* __restrict__ keys)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
  const ulonglong2 a = src[0], b = src[1];      // (node_id, node_offset), (read_id, read_offset)
  uint64_t key = b.x - L.rec_offset;
  key = (key << L.roff_bits) | b.y;
  key = (key << L.node_bits) | id_to_order(L, a.x);
  key = (key << L.noff_bits) | a.y;
  keys[i] = key;
}

__global__ void __launch_bounds__(256)
k_hits_expand(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ n_ptr, KeyLayout L,
              psigpu_hit* __restrict__ out)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_ptr) return;
  uint64_t key = keys[i];
  const uint64_t noff = key & ((1ull << L.noff_bits) - 1ull); key >>= L.noff_bits;
  const uint64_t order = key & ((1ull << L.node_bits) - 1ull); key >>= L.node_bits;
  const uint64_t roff = key & ((1ull << L.roff_bits) - 1ull); key >>= L.roff_bits;
  const uint64_t nid = L.ids_sorted ? L.ids_sorted[order] : L.id_base + order;
  ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + i);
  dst[0] = make_ulonglong2(nid, noff);
  dst[1] = make_ulonglong2(L.rec_offset + key, roff);
}

constexpr uint32_t GROUP_MAX = 32;

// One thread per hit; the thread of the first hit of a group (hits of one seed: equal read id and read
// offset) checks that the group follows the one before it, and orders the group by (node id, node offset).
__global__ void __launch_bounds__(256)
k_hits_fix_groups(psigpu_hit* __restrict__ hits, uint64_t n, const unsigned long long* __restrict__ n_ptr,
                  uint64_t* __restrict__ flag)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_ptr != nullptr) n = min((uint64_t)*n_ptr, n);          // the count is still on the device: n is the capacity
  if (i >= n) return;
  const ulonglong2 b = reinterpret_cast<const ulonglong2*>(hits + i)[1];        // (read_id, read_offset)
  if (i) {
    const ulonglong2 pb = reinterpret_cast<const ulonglong2*>(hits + i - 1)[1];
    if (pb.x == b.x && pb.y == b.y) return;                                      // not the first of its group
    if (pb.x > b.x || (pb.x == b.x && pb.y > b.y)) { *flag = 1; return; }
  }
  uint64_t j = i + 1;
  while (j < n && j - i <= GROUP_MAX) {
    const ulonglong2 nb = reinterpret_cast<const ulonglong2*>(hits + j)[1];
    if (nb.x != b.x || nb.y != b.y) break;
    ++j;
  }
  const uint32_t len = (uint32_t)(j - i);
  if (len == 1) return;
  if (len > GROUP_MAX) { *flag = 1; return; }
  ulonglong2 a[GROUP_MAX];
  bool moved = false;
  for (uint32_t t = 0; t < len; ++t) {                                           // insertion sort
    const ulonglong2 v = reinterpret_cast<const ulonglong2*>(hits + i + t)[0];   // (node_id, node_offset)
    uint32_t p = t;
    while (p && (a[p - 1].x > v.x || (a[p - 1].x == v.x && a[p - 1].y > v.y))) { a[p] = a[p - 1]; --p; }
    a[p] = v;
    moved = moved || p != t;
  }
  for (uint32_t t = 1; t < len; ++t)
    if (a[t].x == a[t - 1].x && a[t].y == a[t - 1].y) { *flag = 1; return; }     // a duplicate: the general path drops it
  if (moved)
    for (uint32_t t = 0; t < len; ++t) reinterpret_cast<ulonglong2*>(hits + i + t)[0] = a[t];
}

inline uint32_t bits_for(uint64_t max_value)      // bits needed to hold 0..max_value (at least 1)
{
  uint32_t b = 1;
  while (b < 64 && (max_value >> b)) ++b;
  return b;
}

}  // namespace

HitSorter::~HitSorter()
{
  if (keys_a) (void)hipFree(keys_a);
  if (keys_b) (void)hipFree(keys_b);
  if (tmp) (void)hipFree(tmp);
}

bool HitSorter::fits(uint64_t n_reads, uint64_t max_read_len, uint64_t n_nodes, uint64_t max_node_len)
{
  return bits_for(n_reads ? n_reads - 1 : 0) + bits_for(max_read_len) + bits_for(n_nodes ? n_nodes - 1 : 0) +
             bits_for(max_node_len) <= 64;
}

int HitSorter::fix_grouped(psigpu_hit* d_hits, uint64_t n, const unsigned long long* d_n, uint64_t* d_flag, void* stream_,
                           std::string* err)
{
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t e = d_n ? hipSuccess : hipMemsetAsync(d_flag, 0, 8, stream);      // (with d_n: the caller's counters, zeroed with them)
  if (e == hipSuccess && n) {
    k_hits_fix_groups<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(d_hits, n, d_n, d_flag);
    e = hipGetLastError();
  }
  if (e != hipSuccess) { *err = std::string("k_hits_fix_groups: ") + hipGetErrorString(e); return PSIGPU_ERR_DEVICE; }
  return PSIGPU_OK;
}

int HitSorter::run(const psigpu_hit* d_in, uint64_t n, uint64_t rec_offset, uint64_t n_reads, uint64_t max_read_len,
                   uint64_t n_nodes, uint64_t max_node_len, bool id_affine, uint64_t id_base,
                   const uint64_t* d_ids_sorted, psigpu_hit* d_out, uint64_t* d_count, void* stream_, std::string* err)
{
  hipStream_t stream = (hipStream_t)stream_;
#define HS_CHK(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      *err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return e_ == hipErrorOutOfMemory ? PSIGPU_ERR_NOMEM : PSIGPU_ERR_DEVICE;             \
    }                                                                                      \
  } while (0)
  KeyLayout L;
  L.noff_bits = bits_for(max_node_len);
  L.node_bits = bits_for(n_nodes ? n_nodes - 1 : 0);
  L.roff_bits = bits_for(max_read_len);
  L.rid_bits = bits_for(n_reads ? n_reads - 1 : 0);
  L.rec_offset = rec_offset; L.id_base = id_base;
  L.ids_sorted = id_affine ? nullptr : d_ids_sorted; L.n_nodes = n_nodes;
  const unsigned end_bit = L.noff_bits + L.node_bits + L.roff_bits + L.rid_bits;
  if (end_bit > 64) { *err = "hit record does not fit a 64-bit sort key"; return PSIGPU_ERR_ARG; }
  if (n == 0) { HS_CHK(hipMemsetAsync(d_count, 0, 8, stream)); return PSIGPU_OK; }
  if (n > cap) {
    if (keys_a) (void)hipFree(keys_a);
    if (keys_b) (void)hipFree(keys_b);
    keys_a = keys_b = nullptr; cap = 0;
    const uint64_t want = n + n / 8 + 1024;
    HS_CHK(hipMalloc(&keys_a, want * 8));
    HS_CHK(hipMalloc(&keys_b, want * 8));
    cap = want;
  }
  uint64_t* ka = (uint64_t*)keys_a;
  uint64_t* kb = (uint64_t*)keys_b;
  size_t need_sort = 0, need_uniq = 0;
  HS_CHK(rocprim::radix_sort_keys(nullptr, need_sort, ka, kb, (size_t)n, 0u, end_bit, stream));
  HS_CHK(rocprim::unique(nullptr, need_uniq, kb, ka, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), stream));
  const size_t need = std::max(need_sort, need_uniq);
  if (need > tmp_cap) {
    if (tmp) (void)hipFree(tmp);
    tmp = nullptr; tmp_cap = 0;
    HS_CHK(hipMalloc(&tmp, need + need / 8 + 256));
    tmp_cap = need + need / 8 + 256;
  }
  const unsigned grid = (unsigned)((n + 255) / 256);
  k_hits_pack<<<grid, 256, 0, stream>>>(d_in, n, L, ka);
  size_t t = tmp_cap;
  HS_CHK(rocprim::radix_sort_keys(tmp, t, ka, kb, (size_t)n, 0u, end_bit, stream));
  t = tmp_cap;
  HS_CHK(rocprim::unique(tmp, t, kb, ka, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), stream));
  k_hits_expand<<<grid, 256, 0, stream>>>(ka, d_count, L, d_out);
  HS_CHK(hipGetLastError());
  return PSIGPU_OK;
#undef HS_CHK
}

}  // namespace psigpu
