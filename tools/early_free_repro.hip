// Does freeing the buffers of a raw HSA engine copy RIGHT AFTER its completion signal reached 0 let ROCr's own
// retirement of that copy write into memory that has meanwhile gone back to the allocator?  (the
// load campaigns caught "one word decremented, a 4-byte zero 736 bytes further" in a Python bytes object while the main
// thread ran nothing but Python -- the shape of a release on freed bookkeeping; the host entry of the library issues such
// copies and used to free their buffers as soon as it had seen the signal.)
//
// Loop: hipHostMalloc a landing buffer, hipMalloc a source, one hsa_amd_memory_async_copy_on_engine device -> host, wait
// for the signal, free BOTH at once (or after `late_us` microseconds: the other arm), then malloc a few hundred small
// blocks of the sizes a runtime's bookkeeping has, fill them with a pattern, give whoever is still writing a moment, and
// check the pattern.  Run several of these beside each other (the events needed a loaded box):
//     hipcc --offload-arch=gfx950 -O2 tools/early_free_repro.hip -o /tmp/efr -lhsa-runtime64
//     for i in $(seq 16); do /tmp/efr 200000 0 & done; wait        # arm "early"
//     for i in $(seq 16); do /tmp/efr 200000 1000000 & done; wait  # arm "late" (frees a second later)
// Prints every block that changed (address, offset, old and new bytes) and a summary line; exit code 1 if any did.
// (Round 5's last 19 seconds of GPU time: 16 processes x 3 000 iterations with the frees at once -- the three that got
// through before the time limit report 0 changed blocks.  Far too few to say anything; run it for minutes.)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <unistd.h>
#include <algorithm>
#include <vector>

static hsa_status_t agents_cb(hsa_agent_t a, void* data)
{
  static_cast<std::vector<hsa_agent_t>*>(data)->push_back(a);
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv)
{
  const long iters = argc > 1 ? atol(argv[1]) : 100000;
  const long late_us = argc > 2 ? atol(argv[2]) : 0;
  const long secs = argc > 3 ? atol(argv[3]) : 0;          // (round 6: stop after this many seconds, 0 = after `iters`)
  const auto t_start = std::chrono::steady_clock::now();
  long done_iters = 0;
  if (hipSetDevice(0) != hipSuccess || hsa_init() != HSA_STATUS_SUCCESS) { fprintf(stderr, "no device\n"); return 2; }
  std::vector<hsa_agent_t> agents;
  hsa_iterate_agents(agents_cb, &agents);
  hsa_agent_t cpu{}, gpu{};
  bool have_cpu = false, have_gpu = false;
  for (hsa_agent_t a : agents) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) continue;
    if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { cpu = a; have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { gpu = a; have_gpu = true; }
  }
  uint32_t mask = 0;
  if (!have_cpu || !have_gpu || hsa_amd_memory_copy_engine_status(cpu, gpu, &mask) != HSA_STATUS_SUCCESS || !mask) {
    fprintf(stderr, "no copy engine\n");
    return 2;
  }
  const uint32_t engine = mask & (~mask + 1);
  hsa_signal_t sig;
  if (hsa_signal_create(0, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return 2;
  struct Held { void* h; void* d; std::chrono::steady_clock::time_point t; };
  std::deque<Held> held;
  const size_t sizes[] = { 48, 64, 96, 128, 192, 256, 512, 880, 1024, 2048 };
  long bad = 0;
  uint64_t rng = 0x9E3779B97F4A7C15ull * (uint64_t)getpid();
  for (long it = 0; it < iters; ++it) {
    if (secs && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(secs)) break;
    ++done_iters;
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    const size_t bytes = 4096 + (rng % 64) * 4096;
    void* h = nullptr; void* d = nullptr;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { fprintf(stderr, "alloc\n"); return 2; }
    hsa_signal_store_relaxed(sig, 1);
    if (hsa_amd_memory_async_copy_on_engine(h, cpu, d, gpu, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)engine, true) != HSA_STATUS_SUCCESS) {
      fprintf(stderr, "copy failed\n");
      return 2;
    }
    while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) >= 1) { }
    if (late_us == 0) { (void)hipHostFree(h); (void)hipFree(d); }
    else held.push_back(Held{ h, d, std::chrono::steady_clock::now() });
    while (!held.empty() && std::chrono::steady_clock::now() - held.front().t > std::chrono::microseconds(late_us)) {
      (void)hipHostFree(held.front().h); (void)hipFree(held.front().d);
      held.pop_front();
    }
    // bait: what the allocator hands out now is what was freed a moment ago
    std::vector<std::pair<unsigned char*, size_t>> bait;
    for (int r = 0; r < 40; ++r)
      for (size_t s : sizes) {
        unsigned char* p = (unsigned char*)malloc(s);
        memset(p, 0x43, s);
        bait.emplace_back(p, s);
      }
    std::this_thread::sleep_for(std::chrono::microseconds(20 + rng % 200));
    for (auto& b : bait) {
      for (size_t i = 0; i < b.second; ++i)
        if (b.first[i] != 0x43) {
          ++bad;
          fprintf(stderr, "iteration %ld: block %p of %zu bytes changed at offset %zu: 0x43 -> 0x%02x (8 bytes there: ", it, (void*)b.first, b.second, i, b.first[i]);
          for (size_t j = i & ~(size_t)7; j < std::min(b.second, (i & ~(size_t)7) + 8); ++j) fprintf(stderr, "%02x", b.first[j]);
          fprintf(stderr, ")\n");
          break;
        }
      free(b.first);
    }
  }
  for (auto& hd : held) { (void)hipHostFree(hd.h); (void)hipFree(hd.d); }
  printf("%ld iterations, frees %s, %ld bait blocks changed\n", done_iters, late_us ? "late" : "at once", bad);
  return bad ? 1 : 0;
}
