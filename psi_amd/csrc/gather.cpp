// The one exchange the path has (BASELINE north_star: "RCCL over xGMI only to gather hit lists"): every rank's
// device-resident hit records into one GPU's HBM, in rank order -- contiguous read ranges per rank, so the ranks'
// sorted arrays in rank order ARE the sorted chunk.  C++ over RCCL: an all-gather of the counts, then one
// point-to-point transfer per rank into the root's buffer inside one group (the shape that fits xGMI's
// point-to-point links; no reduction, no all-to-all).  No counterpart in the reference (single process, host only).
//
// RCCL is loaded on first use (dlopen "librccl.so.1": in a process that already holds one -- PyTorch brings its own
// under the same SONAME -- that one is used), so the library itself does not depend on it: a single-GPU caller never
// pays for loading it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/psi_gpu.h"

namespace {

typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;                         // ncclSuccess == 0
constexpr int NCCL_CHAR = 0, NCCL_UINT64 = 5;     // ncclDataType_t (rccl.h: ncclInt8 / ncclChar = 0, ncclUint64 = 5)

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool ok = false;
};

Rccl& rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) { r.err = std::string("cannot load RCCL: ") + dlerror(); return; }
    bool all = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.h, n); if (!p) { all = false; r.err = std::string("RCCL symbol missing: ") + n; } return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = all;
  });
  return r;
}

thread_local std::string g_comm_err;

}  // namespace

struct psigpu_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void* d_counts = nullptr;       // [world] u64
  void* d_mine = nullptr;         // 1 u64
  void* d_all = nullptr;          // the gathered records (root), grow-only
  size_t all_cap = 0;
  std::string err;
};

extern "C" {

const char* psigpu_comm_last_error(const psigpu_comm* c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

int psigpu_comm_available(void) { return rccl().ok ? 1 : 0; }

int psigpu_comm_unique_id(uint8_t id[PSIGPU_COMM_ID_BYTES])
{
  Rccl& r = rccl();
  if (!r.ok) { g_comm_err = r.err; return PSIGPU_ERR_STATE; }
  ncclUniqueId u;
  ncclResult_t st = r.GetUniqueId(&u);
  if (st != 0) { g_comm_err = std::string("ncclGetUniqueId: ") + r.GetErrorString(st); return PSIGPU_ERR_DEVICE; }
  memcpy(id, u.internal, PSIGPU_COMM_ID_BYTES);
  return PSIGPU_OK;
}

psigpu_comm* psigpu_comm_create(int device, const uint8_t id[PSIGPU_COMM_ID_BYTES], int rank, int world)
{
  Rccl& r = rccl();
  if (!r.ok) { g_comm_err = r.err; return nullptr; }
  if (world < 1 || rank < 0 || rank >= world || !id) { g_comm_err = "bad rank / world"; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return nullptr; }
  psigpu_comm* c = new psigpu_comm;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, PSIGPU_COMM_ID_BYTES);
  ncclResult_t st = r.CommInitRank(&c->comm, world, u, rank);          // collective: every rank of the world calls it
  if (st != 0) { g_comm_err = std::string("ncclCommInitRank: ") + r.GetErrorString(st); delete c; return nullptr; }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&c->d_counts, (size_t)world * 8) != hipSuccess || hipMalloc(&c->d_mine, 8) != hipSuccess) {
    g_comm_err = "cannot allocate the communicator's buffers";
    psigpu_comm_destroy(c);
    return nullptr;
  }
  return c;
}

void psigpu_comm_destroy(psigpu_comm* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (void* p : { c->d_counts, c->d_mine, c->d_all }) if (p) (void)hipFree(p);
  delete c;
}

int psigpu_gather_hits(psigpu_comm* c, const psigpu_hit* d_hits, uint64_t n, int root, const psigpu_hit** d_all, uint64_t* n_all,
                       uint64_t* counts /* [world], may be NULL */)
{
  if (!c || root < 0 || root >= c->world || (n && !d_hits)) return PSIGPU_ERR_ARG;
  Rccl& r = rccl();
  if (!c->comm) { c->err = "the communicator was aborted by an earlier failure"; return PSIGPU_ERR_STATE; }
  // A failure of the collective calls themselves leaves the communicator in an unknown state on this rank and the
  // peers possibly blocked in theirs: abort it (ncclCommAbort makes the peers' pending operations fail instead of
  // hang), every later gather on it returns PSIGPU_ERR_STATE.
  auto abort_comm = [&] { if (c->comm) { (void)r.CommAbort(c->comm); c->comm = nullptr; } };
  auto fail = [&](const char* what, ncclResult_t st) { c->err = std::string(what) + ": " + r.GetErrorString(st); abort_comm(); return PSIGPU_ERR_DEVICE; };
  auto hfail = [&](const char* what, hipError_t e) { c->err = std::string(what) + ": " + hipGetErrorString(e); abort_comm(); return PSIGPU_ERR_DEVICE; };
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hfail("hipSetDevice", e);
  // 1. everybody learns everybody's count
  if ((e = hipMemcpyAsync(c->d_mine, &n, 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return hfail("hipMemcpyAsync", e);
  ncclResult_t st = r.AllGather(c->d_mine, c->d_counts, 1, NCCL_UINT64, c->comm, c->stream);
  if (st != 0) return fail("ncclAllGather", st);
  std::vector<uint64_t> cnt((size_t)c->world);
  if ((e = hipMemcpyAsync(cnt.data(), c->d_counts, (size_t)c->world * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return hfail("hipMemcpyAsync", e);
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return hfail("hipStreamSynchronize", e);
  uint64_t total = 0;
  for (uint64_t x : cnt) total += x;
  if (counts) memcpy(counts, cnt.data(), (size_t)c->world * 8);
  // 2. the root makes room; whether it could is agreed on by everybody BEFORE any transfer is posted (a root that
  // returned early used to leave the senders blocked in ncclSend with nothing to time them out): a second all-gather of
  // one status word per rank -- 0 = ready -- and every rank returns the same error when one is not
  uint64_t my_status = 0;
  std::string local_err;
  if (c->rank == root) {
    const size_t need = (size_t)(total + 1) * sizeof(psigpu_hit);
    if (need > c->all_cap) {
      if (c->d_all) (void)hipFree(c->d_all);
      c->d_all = nullptr; c->all_cap = 0;
      if ((e = hipMalloc(&c->d_all, need + need / 8)) != hipSuccess) {
        (void)hipGetLastError();
        my_status = 1; local_err = std::string("hipMalloc (gathered hits): ") + hipGetErrorString(e);
      } else c->all_cap = need + need / 8;
    }
  }
  if ((e = hipMemcpyAsync(c->d_mine, &my_status, 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return hfail("hipMemcpyAsync", e);
  if ((st = r.AllGather(c->d_mine, c->d_counts, 1, NCCL_UINT64, c->comm, c->stream)) != 0) return fail("ncclAllGather (status)", st);
  std::vector<uint64_t> status((size_t)c->world);
  if ((e = hipMemcpyAsync(status.data(), c->d_counts, (size_t)c->world * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return hfail("hipMemcpyAsync", e);
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return hfail("hipStreamSynchronize", e);
  for (int p = 0; p < c->world; ++p)
    if (status[(size_t)p]) {
      c->err = p == c->rank ? local_err : "rank " + std::to_string(p) + " could not take part in the gather (out of device memory)";
      return PSIGPU_ERR_NOMEM;                 // on every rank alike; the communicator stays usable
    }
  // 3. payloads: point to point into the root's buffer, in rank order, one group
  if ((st = r.GroupStart()) != 0) return fail("ncclGroupStart", st);
  if (c->rank == root) {
    uint64_t at = 0;
    for (int p = 0; p < c->world; ++p) {
      char* dst = (char*)c->d_all + at * sizeof(psigpu_hit);
      if (p == root) {
        if (cnt[p] && (e = hipMemcpyAsync(dst, d_hits, cnt[p] * sizeof(psigpu_hit), hipMemcpyDeviceToDevice, c->stream)) != hipSuccess) {
          (void)r.GroupEnd();
          return hfail("hipMemcpyAsync (own hits)", e);
        }
      } else if (cnt[p]) {
        if ((st = r.Recv(dst, cnt[p] * sizeof(psigpu_hit), NCCL_CHAR, p, c->comm, c->stream)) != 0) { (void)r.GroupEnd(); return fail("ncclRecv", st); }
      }
      at += cnt[p];
    }
  } else if (n) {
    if ((st = r.Send(d_hits, n * sizeof(psigpu_hit), NCCL_CHAR, root, c->comm, c->stream)) != 0) { (void)r.GroupEnd(); return fail("ncclSend", st); }
  }
  if ((st = r.GroupEnd()) != 0) return fail("ncclGroupEnd", st);
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return hfail("hipStreamSynchronize", e);
  if (d_all) *d_all = c->rank == root ? (const psigpu_hit*)c->d_all : nullptr;
  if (n_all) *n_all = c->rank == root ? total : 0;
  return PSIGPU_OK;
}

}  // extern "C"
