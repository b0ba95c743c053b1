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
  uint32_t bytes = 0;                          // 8 or 16 (0: no wire records)
  uint32_t noff_bits = 0, node_bits = 0, roff_bits = 0;      // W8; the read id has the remaining 64 - sum bits
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

