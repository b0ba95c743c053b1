// Do a PCIe copy-out kernel on one stream and compute kernels on another overlap on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
int main()
{
  const size_t piece = 24u << 20, dev_bytes = 256u << 20;
  char *h_out, *h_in, *d_a, *d_b, *d_c;
  CK(hipHostMalloc((void**)&h_out, piece, hipHostMallocMapped)); CK(hipHostMalloc((void**)&h_in, piece, hipHostMallocMapped));
  CK(hipMalloc((void**)&d_a, dev_bytes)); CK(hipMalloc((void**)&d_b, dev_bytes)); CK(hipMalloc((void**)&d_c, piece));
  hipStream_t sc, so, si;
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&so, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&si, hipStreamNonBlocking));
  hipEvent_t e[8];
  for (auto& x : e) CK(hipEventCreate(&x));
  for (int grid : { 64, 256, 1024 }) {
    for (int variant = 0; variant < 3; ++variant) {       // 0: compute alone, 1: copy-out alone, 2: both, 3 = + SDMA H2D
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        double t0 = now();
        CK(hipEventRecord(e[0], sc)); CK(hipEventRecord(e[2], so));
        if (variant != 1) for (int j = 0; j < 8; ++j) k_copy16<<<2048, 256, 0, sc>>>((const uint4*)d_a, (uint4*)d_b, dev_bytes / 16 / 4);
        if (variant != 0) k_copy16<<<grid, 256, 0, so>>>((const uint4*)d_c, (uint4*)h_out, piece / 16);
        if (variant == 2) CK(hipMemcpyAsync(d_c, h_in, piece, hipMemcpyHostToDevice, si));
        CK(hipEventRecord(e[1], sc)); CK(hipEventRecord(e[3], so));
        CK(hipStreamSynchronize(sc)); CK(hipStreamSynchronize(so)); CK(hipStreamSynchronize(si));
        double dt = now() - t0;
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&b, e[2], e[3]);
        if (rep == 2) printf("grid %4d variant %d: wall %.3f ms, compute stream %.3f ms, copy-out stream %.3f ms\n", grid, variant, dt * 1e3, a, b);
      }
    }
  }
  return 0;
}
