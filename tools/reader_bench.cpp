// Reads a FASTQ through psi::SeqStreamIn (parallel reader) and times parsing and packing apart: tools/reader_bench FILE CHUNK NAMES(0/1)
//   g++ -O3 -std=c++17 -Ipsi_amd/include -Iinclude tools/reader_bench.cpp -o /tmp/reader_bench -Lpsi_amd -lpsi_gpu -lz -lpthread -Wl,-rpath,$PWD/psi_amd
#include <chrono>
#include <cstdio>
#include "psi/sequence.hpp"
int main(int argc, char** argv) {
  uint64_t chunk = argc > 2 ? atoll(argv[2]) : 0;
  bool names = argc > 3 ? atoi(argv[3]) : 1;
  for (int rep = 0; rep < 3; ++rep) {
    psi::SeqStreamIn iss(argv[1]);
    psi::Records r; r.keep_names = names;
    auto t0 = std::chrono::steady_clock::now();
    uint64_t tot = 0; double tp = 0;
    while (true) {
      int f = iss.read_chunk_fast(r, chunk);
      if (f <= 0) break;
      auto t1 = std::chrono::steady_clock::now();
      r.pack();
      tp += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      tot += r.size();
    }
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%llu reads in %.3f s (pack %.3f) = %.2f M reads/s parse-only %.2f M/s\n", (unsigned long long)tot, s, tp, tot / s / 1e6, tot / (s - tp) / 1e6);
  }
}
