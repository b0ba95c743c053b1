// What separates k_kmer_probe's 30 G seeds/s from the 47-53 G random sectors/s of tools/rand_sector.hip?  The probe's SHAPE,
// piece by piece, on a table of the chr22 size (2.8 GB of 16-byte slots), 7 M keys per launch, 8192 waves:
//   v0  key stream -> one 16-byte load at a precomputed random slot -> 8-byte result stream      (the probe without its logic)
//   v1  v0 with the slot computed from the key by the probe's hash (mix64 + umul64hi)
//   v2  v1 + a second, dependent look in the same 64-byte sector for a third of the lanes        (the re-probe chain)
//   v3  v0 with all four slots of the sector loaded at once (4 independent 16-byte loads)         (sector-at-once)
//   v4  v0 with 28 M keys per launch (four times the work: ramp and tail of a 0.2-ms kernel)
//   v5  v0, 2 keys per lane in flight
// hipcc --offload-arch=gfx950 -O3 tools/probe_shape.hip -o /tmp/probe_shape && /tmp/probe_shape
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int V>
__global__ void __launch_bounds__(256) k_probe(const uint4* __restrict__ t, uint64_t n_slots, const uint64_t* __restrict__ keys,
                                               uint64_t n, uint32_t per_wave, uint64_t* __restrict__ out)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t s0 = wave * per_wave, s1 = min(n, s0 + per_wave);
  if (V == 5) {
    for (uint64_t base = s0; base < s1; base += 128) {
      const uint64_t a = base + lane, b = base + 64 + lane;
      uint64_t ka = a < s1 ? keys[a] : 0, kb = b < s1 ? keys[b] : 0;
      uint4 va = t[ka % n_slots], vb = t[kb % n_slots];
      if (a < s1) out[a] = (uint64_t)va.x | ((uint64_t)va.w << 32);
      if (b < s1) out[b] = (uint64_t)vb.x | ((uint64_t)vb.w << 32);
    }
    return;
  }
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t seed = base + lane;
    if (seed >= s1) continue;
    const uint64_t key = keys[seed];
    uint64_t h = (V == 0 || V == 3 || V == 4) ? key % n_slots : __umul64hi(mix64(key), n_slots);
    uint64_t r;
    if (V == 3) {
      const uint4* s = t + (h & ~3ull);
      uint4 a = s[0], b = s[1], c = s[2], d = s[3];
      r = (uint64_t)(a.x ^ b.x ^ c.x ^ d.x) | ((uint64_t)(a.w + b.w + c.w + d.w) << 32);
    } else {
      uint4 v = t[h];
      if (V == 2 && (key & 3) == 0) {          // (a dependent second look: the address depends on what came back)
        uint4 w = t[(h & ~3ull) | ((h + 1 + (v.x & 1)) & 3ull)];
        v.x ^= w.x; v.w += w.w;
      }
      r = (uint64_t)v.x | ((uint64_t)v.w << 32);
    }
    out[seed] = r;
  }
}

// the emit kernel's shape: 8 bytes in, one 32-byte record out per item -- w0: every lane stores its own record (two 16-byte
// stores 32 bytes apart: an instruction covers half of every 64-byte line it touches); w1: the records of 32 lanes
// transposed by shuffles so that an instruction stores 1 KB without holes
template <int W>
__global__ void __launch_bounds__(256) k_write(const uint64_t* __restrict__ in, uint64_t n, uint32_t per_wave, ulonglong2* __restrict__ out)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t s0 = wave * per_wave, s1 = min(n, s0 + per_wave);
  for (uint64_t base = s0; base < s1; base += 64) {
    const uint64_t item = base + lane;
    const uint64_t r = item < s1 ? in[item] : 0;
    const uint64_t nid = r & 0xFFFFFFFFu, noff = (r >> 32) & 0xFFFFFFFu, rid = item / 7, roff = (item % 7) * 21;
    if (W == 0) {
      if (item < s1) { out[2 * item] = make_ulonglong2(nid, noff); out[2 * item + 1] = make_ulonglong2(rid, roff); }
    } else {
      if (base + 64 <= s1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int src = 32 * h + (int)(lane >> 1);
          const uint32_t a0 = __shfl((int)(uint32_t)nid, src), a1 = __shfl((int)(uint32_t)(nid >> 32), src), a2 = __shfl((int)(uint32_t)noff, src);
          const uint32_t b0 = __shfl((int)(uint32_t)rid, src), b1 = __shfl((int)(uint32_t)(rid >> 32), src), b2 = __shfl((int)(uint32_t)roff, src);
          const bool second = lane & 1;
          const uint64_t x = second ? ((uint64_t)b0 | ((uint64_t)b1 << 32)) : ((uint64_t)a0 | ((uint64_t)a1 << 32));
          const uint64_t y = second ? b2 : a2;
          out[2 * base + 64 * h + lane] = make_ulonglong2(x, y);
        }
      } else if (item < s1) { out[2 * item] = make_ulonglong2(nid, noff); out[2 * item + 1] = make_ulonglong2(rid, roff); }
    }
  }
}

template <int W>
static void run_w(const char* what, const uint64_t* in, uint64_t n, ulonglong2* out)
{
  const uint64_t n_waves = 8192;
  const uint32_t per_wave = (uint32_t)(((n + n_waves - 1) / n_waves + 63) / 64 * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_write<W><<<n_waves / 4, 256>>>(in, n, per_wave, out);
  const int reps = 20;
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k_write<W><<<n_waves / 4, 256>>>(in, n, per_wave, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  printf("%-64s %7.3f ms  %6.2f TB/s\n", what, ms, n * 40.0 / ms / 1e9);
}

template <int V>
static void run(const char* what, const uint4* t, uint64_t n_slots, const uint64_t* keys, uint64_t n, uint64_t* out)
{
  const uint64_t n_waves = 8192;
  const uint32_t per_wave = (uint32_t)(((n + n_waves - 1) / n_waves + 63) / 64 * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_probe<V><<<n_waves / 4, 256>>>(t, n_slots, keys, n, per_wave, out);
  const int reps = 20;
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) k_probe<V><<<n_waves / 4, 256>>>(t, n_slots, keys, n, per_wave, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  printf("%-64s %7.3f ms  %6.2f G keys/s\n", what, ms, n / ms / 1e6);
}

int main()
{
  const uint64_t n_slots = (2800ull << 20) / 16 / 4 * 4;
  const uint64_t n = 7000000, n4 = 4 * n;
  uint4* t; uint64_t *keys, *out;
  if (hipMalloc(&t, n_slots * 16) != hipSuccess || hipMalloc(&keys, n4 * 8) != hipSuccess || hipMalloc(&out, n4 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(t, 1, n_slots * 16);
  std::vector<uint64_t> hk(n4);
  uint64_t x = 88172645463325252ull;
  for (auto& k : hk) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k = x >> 2; }
  hipMemcpy(keys, hk.data(), n4 * 8, hipMemcpyHostToDevice);
  run<0>("v0 key stream, one 16-B load per key, 8-B result", t, n_slots, keys, n, out);
  run<1>("v1 v0 + the probe's hash", t, n_slots, keys, n, out);
  run<2>("v2 v1 + a dependent second look in the sector for 1/4 of the lanes", t, n_slots, keys, n, out);
  run<3>("v3 v0 with the whole sector (4 x 16 B) loaded at once", t, n_slots, keys, n, out);
  run<4>("v4 v0 with 28 M keys per launch", t, n_slots, keys, n4, out);
  run<5>("v5 v0 with two keys per lane in flight", t, n_slots, keys, n, out);
  run<0>("v0 again", t, n_slots, keys, n, out);
  ulonglong2* recs;
  if (hipMalloc(&recs, n * 32) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run_w<0>("w0 8 B in, 32-B record out, every lane its own record", keys, n, recs);
  run_w<1>("w1 the same, records transposed: 1 KB per store instruction", keys, n, recs);
  run_w<0>("w0 again", keys, n, recs);
  return 0;
}
