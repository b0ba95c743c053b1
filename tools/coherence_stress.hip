// Does a kernel ever see a STALE line of ordinary device memory that an earlier kernel of the same stream wrote -- when
// several processes share the GPU?  (Round-3's load campaign saw one record missing / two extra in ~30 000 finder runs
// under eight processes per GPU, never alone; both symptoms are what a stale read of the call's counters would give:
// tools/stress.py is the product-level campaign, this is the pattern in isolation.)
//
// The library's hand-back of a call's counters, reduced to its memory pattern:
//   kernel A (many workgroups, any XCD): every workgroup adds to striped counters with device-scope atomics and the last
//            thread of the grid stores the call's serial number into a plain word (ordinary hipMalloc memory);
//   kernel B (ONE workgroup, like k_publish): copies the counter block to MAPPED pinned host memory with plain loads;
//   host: hipStreamSynchronize, then checks every word against the serial it passed in.
// The block is rewritten every iteration with a new serial, on a non-blocking stream, tens of thousands of times per second.
// A mismatch = the copy kernel read a value of an earlier iteration.   hipcc --offload-arch=gfx950 -O2 -o coherence_stress ...
// usage: coherence_stress ITERATIONS [WORKGROUPS]      (run N of them at once to share the GPU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

constexpr int LINES = 32;                       // 128-byte lines, one word used in each
struct alignas(128) Line { unsigned long long v; char pad[120]; };

__global__ void k_zero(Line* c) { if (threadIdx.x < LINES + 2) c[threadIdx.x].v = 0; }

__global__ void __launch_bounds__(256) k_work(Line* c, unsigned long long serial, unsigned long long* done)
{
  if (threadIdx.x == 0) atomicAdd(&c[blockIdx.x % LINES].v, serial);                       // striped statistics counter
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = atomicAdd(done, 1ull);
    if (t + 1 == gridDim.x) { c[LINES].v = serial; c[LINES + 1].v = serial * 3 + 1; *done = 0; }   // plain stores by the grid's last workgroup
  }
}

__global__ void __launch_bounds__(256) k_publish(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16)
{
  for (uint32_t i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
}

int main(int argc, char** argv)
{
  const long iters = argc > 1 ? atol(argv[1]) : 100000;
  const int wgs = argc > 2 ? atoi(argv[2]) : 96;
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Line* c; unsigned long long* done; void* h; void* h_dev;
  CHK(hipMalloc(&c, sizeof(Line) * (LINES + 2)));
  CHK(hipMalloc(&done, 8));
  CHK(hipMemset(done, 0, 8));
  CHK(hipHostMalloc(&h, sizeof(Line) * (LINES + 2), hipHostMallocMapped));
  CHK(hipHostGetDevicePointer(&h_dev, h, 0));
  const Line* hl = (const Line*)h;
  long bad = 0;
  for (long it = 1; it <= iters; ++it) {
    const unsigned long long serial = (unsigned long long)it;
    k_zero<<<1, 64, 0, s>>>(c);
    k_work<<<wgs, 256, 0, s>>>(c, serial, done);
    k_publish<<<1, 256, 0, s>>>((const uint4*)c, (uint4*)h_dev, (uint32_t)(sizeof(Line) * (LINES + 2) / 16));
    CHK(hipStreamSynchronize(s));
    unsigned long long sum = 0;
    for (int i = 0; i < LINES; ++i) sum += hl[i].v;
    const bool ok = sum == serial * (unsigned long long)wgs && hl[LINES].v == serial && hl[LINES + 1].v == serial * 3 + 1;
    if (!ok) {
      if (++bad <= 5) fprintf(stderr, "STALE at iteration %ld: sum %llu (want %llu), serial word %llu, %llu\n", it, sum,
                              serial * (unsigned long long)wgs, hl[LINES].v, hl[LINES + 1].v);
    }
  }
  printf("{\"iterations\": %ld, \"workgroups\": %d, \"stale\": %ld}\n", iters, wgs, bad);
  return bad ? 1 : 0;
}
