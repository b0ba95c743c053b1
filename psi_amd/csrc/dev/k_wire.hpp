// wire records of the host entry; the random-access rate measured in place -- part of the one translation unit device.hip (included there, in order; not a header of its own).
// ------------------------------------------------------------------------------------
// The host entry's wire format.  A hit record is 4 x u64 (psi::Seed<> as psikt writes it), but of its 32 bytes
// only about 12 carry information: the link out of the device is the bound of the host entry (224 MB of
// records against 158 MB of reads per 1 M-read chunk), so the records cross it as 4 x u32 -- node id minus the
// graph's first id (ids that are rank + constant), node offset, read id minus the sub-batch's first, read offset
// -- and host threads widen them into the caller's 32-byte records while the next sub-batch is in flight.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_hits_wire16(const psigpu_hit* __restrict__ hits, const unsigned long long* __restrict__ n_a, const unsigned long long* __restrict__ n_b,
              uint64_t n_fixed, uint64_t cap, uint64_t id_base, uint64_t rec_base, uint4* __restrict__ out)
{
  // the number of hits: on the device (n_a [+ n_b]) when the host does not know it yet, else n_fixed
  const uint64_t n = min(n_a ? (uint64_t)*n_a + (n_b ? (uint64_t)*n_b : 0ull) : n_fixed, cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
    const ulonglong2 a = src[0], b = src[1];
    out[i] = make_uint4((uint32_t)(a.x - id_base), (uint32_t)a.y, (uint32_t)(b.x - rec_base), (uint32_t)b.y);
  }
}

// Round 4: 8 bytes per record.  The four fields of a hit need far fewer than 64 bits together -- the device sorter
// already packs them into one 64-bit key (hits_gpu.hip) -- so a sub-batch's records cross the link as ONE u64 each:
//     [ read id - the sub-batch's first | read offset | node id - the graph's first id | node offset ]
// with the node fields sized by the graph (bits for its largest node length and its node count), and the read offset
// given every bit the read id of the sub-batch leaves (chr22-like: 17 + 19 + 22 + 6).  The kernel CHECKS that every
// field fits -- the longest read is not known to the host when the kernel is queued -- and raises a flag in mapped host
// memory when one does not; the host entry then makes 16-byte records of that sub-batch instead.  56 MB instead of
// 112 (round 3) or 224 (rounds 1-2) per 1 M-read chunk.
struct WireFmt {
  uint32_t bytes = 0;                          // 5 / 6 / 7 (packed, below), 8 or 16 (0: no wire records)
  uint32_t noff_bits = 0, node_bits = 0, roff_bits = 0;      // W8 and packed; the read id has the remaining 8 x bytes - sum bits
  uint32_t roff_div = 1;                       // packed: the read offset is carried in units of this (the seed distance)
};

__global__ void __launch_bounds__(256)
k_hits_wire8(const psigpu_hit* __restrict__ hits, const unsigned long long* __restrict__ n_a, const unsigned long long* __restrict__ n_b,
             uint64_t n_fixed, uint64_t cap, uint64_t id_base, uint64_t rec_base, WireFmt f, uint64_t* __restrict__ out,
             unsigned long long* __restrict__ overflow /* mapped host memory */)
{
  const uint64_t n = min(n_a ? (uint64_t)*n_a + (n_b ? (uint64_t)*n_b : 0ull) : n_fixed, cap);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint32_t rid_bits = 64 - f.noff_bits - f.node_bits - f.roff_bits;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
    const ulonglong2 a = src[0], b = src[1];
    const uint64_t node = a.x - id_base, rid = b.x - rec_base;
    bad = bad || (a.y >> f.noff_bits) || (node >> f.node_bits) || (b.y >> f.roff_bits) || (rid_bits < 64 && (rid >> rid_bits));
    out[i] = ((((rid << f.roff_bits) | b.y) << f.node_bits | node) << f.noff_bits) | a.y;
  }
  if (__any(bad) && lane_id() == 0) *overflow = 1ull;
}

// Round 5: 5 to 7 bytes per record ("packed").  Of the 64 bits of an 8-byte key about 17 + 19 say which read of the
// sub-batch and where in it; but the records leave the device in read order (sorted, or as the seeds were), so inside a
// BLOCK of WP_BLOCK consecutive records the read id is the block's first plus a small number, and the offset in the read is
// a multiple of the seed distance.  A block crosses the link as
//     [ u32 read id of its first record - the sub-batch's first | 4 bytes unused ] [ WP_BLOCK records of `bytes` bytes each ]
//     record (little endian, `bytes` bytes):  read id - the block's first | read offset / seed distance | node | node offset
// chr22-like: 9 + 3 + 22 + 6 bits = 5 bytes (35 MB per 1 M-read chunk instead of 56).  The kernel checks every field of every
// record and raises a flag in mapped host memory when one does not fit: overflow[0] -- a wider record may do (the host
// entry goes to bytes + 1 for the context) -- and overflow[1] as well when it would not (records not in read order, an
// offset that is no multiple of the seed distance: 8-byte keys from then on).
constexpr uint32_t WP_BLOCK = 256;
__host__ __device__ inline uint64_t wirep_block_stride(uint32_t bytes) { return 8ull + (uint64_t)WP_BLOCK * bytes; }
// bytes of n packed records (whole 8-byte words)
static inline uint64_t wire_bytes(const WireFmt& f, uint64_t n)
{
  if (f.bytes >= 8 || f.bytes == 0) return n * f.bytes;
  const uint64_t full = n / WP_BLOCK, rem = n % WP_BLOCK;
  return full * wirep_block_stride(f.bytes) + (rem ? 8 + (rem * f.bytes + 7) / 8 * 8 : 0);
}

__global__ void __launch_bounds__(256)
k_hits_wirep(const psigpu_hit* __restrict__ hits, const unsigned long long* __restrict__ n_a, const unsigned long long* __restrict__ n_b,
             uint64_t n_fixed, uint64_t cap, uint64_t id_base, uint64_t rec_base, WireFmt f, uint64_t* __restrict__ out,
             unsigned long long* __restrict__ overflow /* mapped host memory */)
{
  // a workgroup per four blocks: a record per thread and block, its key into LDS, then the blocks' words out whole
  __shared__ uint64_t keys[4][WP_BLOCK + 2];
  const uint64_t n = min(n_a ? (uint64_t)*n_a + (n_b ? (uint64_t)*n_b : 0ull) : n_fixed, cap);
  const uint32_t W = 8 * f.bytes, rel_bits = W - f.noff_bits - f.node_bits - f.roff_bits;
  const uint64_t stride_w = wirep_block_stride(f.bytes) / 8;      // words per block (8 + 256 x bytes is a multiple of 8)
  const uint32_t words = WP_BLOCK * f.bytes / 8;
  uint32_t bad = 0;
  if (threadIdx.x < 8) keys[threadIdx.x >> 1][WP_BLOCK + (threadIdx.x & 1)] = 0;
  for (uint64_t g0 = (uint64_t)blockIdx.x * 4 * WP_BLOCK; g0 < n; g0 += (uint64_t)gridDim.x * 4 * WP_BLOCK) {
    uint64_t base[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
      const uint64_t i0 = g0 + (uint64_t)b * WP_BLOCK, i = i0 + threadIdx.x;
      base[b] = 0;
      uint64_t key = 0;
      if (i < n) {
        base[b] = hits[i0].read_id - rec_base;      // (one address for the whole workgroup)
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(hits + i);
        const ulonglong2 a = src[0], r = src[1];
        const uint64_t node = a.x - id_base, rid = r.x - rec_base, ro = r.y / f.roff_div, rel = rid - base[b];
        if (rid < base[b] || ro * f.roff_div != r.y || (a.y >> f.noff_bits) || (node >> f.node_bits) || (base[b] >> 32)) bad |= 2u;
        else if ((ro >> f.roff_bits) || (rel >> rel_bits)) bad |= 1u;
        key = ((((rel << f.roff_bits) | ro) << f.node_bits | node) << f.noff_bits) | a.y;
        key &= W < 64 ? (1ull << W) - 1 : ~0ull;
      }
      keys[b][threadIdx.x] = key;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
      const uint64_t i0 = g0 + (uint64_t)b * WP_BLOCK;
      if (i0 >= n) break;
      const uint32_t have = (uint32_t)min((uint64_t)WP_BLOCK, n - i0);
      const uint32_t nw = (have * f.bytes + 7) / 8;              // words of this block that carry records
      uint64_t* dst = out + (i0 / WP_BLOCK) * stride_w;
      if (threadIdx.x == 0) dst[0] = base[b];
      for (uint32_t w = threadIdx.x; w < nw && w < words; w += 256) {
        // bits [64 w, 64 w + 64) of the records laid end to end, W bits each
        uint32_t r = 64 * w / W;
        const uint32_t o = 64 * w - r * W;
        uint64_t word = keys[b][r] >> o;
        for (uint32_t filled = W - o; filled < 64; filled += W) word |= keys[b][++r] << filled;
        dst[1 + w] = word;
      }
    }
    __syncthreads();
  }
  // (two words, plain stores: overflow[0] = some field did not fit, overflow[1] = ... and a wider packed record would not help)
  if (bad) overflow[0] = 1ull;
  if (bad & 2u) overflow[1] = 1ull;
}

// ------------------------------------------------------------------------------------
// The part's random-access rate, measured in place (psigpu_measure_random_loads): what the probe of the k-mer
// table (one divergent 16-byte load per lane) and the LF / locate kernels (one 64-byte sector per quad) are
// bounded by.  QUAD = false: every lane loads 16 bytes from a sector of its own; QUAD = true: the four lanes
// of a quad load the four 16-byte pieces of one sector.  Addresses come from a hash of the thread and the
// iteration; `iters` loads per thread, each depending on nothing.
// ------------------------------------------------------------------------------------
template <bool QUAD>
__global__ void __launch_bounds__(256) k_rand_loads(const uint4* __restrict__ t, uint64_t n_sectors, uint32_t iters, uint32_t* out)
{
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t x = (QUAD ? (tid >> 2) : tid) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < iters; ++i) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
    const uint64_t sct = __umul64hi(x, n_sectors);
    const uint4 a = t[sct * 4 + (QUAD ? (tid & 3) : ((x >> 5) & 3))];
    acc ^= a.x + a.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

