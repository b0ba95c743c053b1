// MI355X (gfx950) seed-finding kernels and the device half of the C ABI.
//
// One call of psigpu_find_seeds* = one chunk of psikt's loop (reference src/psikt.cpp:195-204):
//
//   K0  k_seed_scan_* / k_seed_pack         seeding()                 include/psi/sequence.hpp:1688-1745
//       k_table_insert                      index_reads()             include/psi/seed_finder.hpp:1089-1097
//   K1  k_fm_search[_direct]                kmer_exact_matches descent include/psi/index_iter.hpp:835-841
//                                           -> Iter::go_down           include/psi/fmindex.hpp:851-869
//       k_kmer_probe                        the same result from the k-mer table (PSIGPU_MODE_KMER_TABLE)
//   K2  k_fm_locate_direct / k_fm_walk      get_occurrences + mapping  include/psi/fmindex.hpp:734-777,
//                                                                      include/psi/pathindex.hpp:378-416
//   K4  k_traverse<false>                   TraverserBFS::run          include/psi/traverser_bfs.hpp:72-161
//       k_traverse<true>                    the same walks enumerated once per index (the tables)
//   K5  emission: scan-placed records in K2, private chunks in K4 (callbacks at index_iter.hpp:676,
//       traverser_bfs.hpp:109)
//
// ONE translation unit, kept as the files of dev/ by what they hold (round 5; it was a single 7000-line file): the kernels
// by family (dev/k_*.hpp), the context (dev/ctx.hpp), and the C ABI by role (dev/*.inc: context and loaders, tables,
// the per-chunk pipeline, the device-resident entries, the host entry).  They are included below in dependency order and
// share the anonymous namespace / the extern "C" block they are included in.
//
// Integer rank / popcount / compare work, HBM-latency and -bandwidth bound; no MFMA.
// Wavefronts are 64 lanes.  A "quad" is 4 adjacent lanes that fetch one 64-byte rank block
// as 4 x 16 B (one coalesced sector) and combine partial popcounts with DPP quad permutes.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <cctype>
#include <execinfo.h>
#include <pthread.h>
#include <sched.h>
#include <memory>
#include <signal.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "host.hpp"

using namespace psigpu;

namespace {
#include "dev/types.hpp"
#include "dev/k_seed.hpp"
#include "dev/k_fm_search.hpp"
#include "dev/k_fm_sweep.hpp"
#include "dev/k_emit_common.hpp"
#include "dev/k_locus_table.hpp"
#include "dev/k_kmer_table.hpp"
#include "dev/k_locate.hpp"
#include "dev/k_kmer_step.hpp"
#include "dev/k_traverse.hpp"
#include "dev/k_mem.hpp"
#include "dev/k_wire.hpp"
#include "dev/devbuf.hpp"
}  // namespace

#include "dev/ctx.hpp"

extern "C" {
#include "dev/api_context.inc"
#include "dev/tables.inc"
#include "dev/pipeline.inc"
#include "dev/api_query.inc"
}  // extern "C"

#include "dev/host_workers.hpp"

extern "C" {
#include "dev/host_entry.inc"
}  // extern "C"
