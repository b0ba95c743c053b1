// What the host link delivers on this box: pinned H2D, pinned D2H, and both at once (full duplex),
// in pieces of the size the host entry point's sub-batches use.  hipcc -O2 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int pcie_rate_run()
{
  const size_t total = 256u << 20;
  char *h_in, *h_out, *d_in, *d_out;
  CK(hipHostMalloc((void**)&h_in, total)); CK(hipHostMalloc((void**)&h_out, total));
  CK(hipMalloc((void**)&d_in, total)); CK(hipMalloc((void**)&d_out, total));
  for (size_t i = 0; i < total; i += 4096) { h_in[i] = 1; h_out[i] = 1; }
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (size_t piece : { (size_t)16 << 20, (size_t)256 << 20 }) {
    for (int mode = 0; mode < 6; ++mode) {       // 0: H2D, 1: D2H, 2: both; 3: D2H by kernel; 4: H2D by SDMA + D2H by kernel; 5: H2D by kernel
      double best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (size_t o = 0; o < total; o += piece) {
          if (mode == 0 || mode == 2 || mode == 4) CK(hipMemcpyAsync(d_in + o, h_in + o, piece, hipMemcpyHostToDevice, a));
          if (mode == 1 || mode == 2) CK(hipMemcpyAsync(h_out + o, d_out + o, piece, hipMemcpyDeviceToHost, b));
          if (mode == 3 || mode == 4) k_copy16<<<256, 256, 0, b>>>((const uint4*)(d_out + o), (uint4*)(h_out + o), piece / 16);
          if (mode == 5) k_copy16<<<256, 256, 0, a>>>((const uint4*)(h_in + o), (uint4*)(d_in + o), piece / 16);
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
        best = std::min(best, now() - t0);
      }
      printf("piece %4zu MiB  %-5s  %.2f ms for 256 MiB%s = %.1f GB/s per direction\n", piece >> 20,
             mode == 0 ? "H2D" : mode == 1 ? "D2H" : mode == 2 ? "both" : mode == 3 ? "D2Hk" : mode == 4 ? "H2D+D2Hk" : "H2Dk", best * 1e3,
             (mode == 2 || mode == 4) ? " each way" : "", total / best / 1e9);
    }
  }
  return 0;
}

int main() { return pcie_rate_run(); }
