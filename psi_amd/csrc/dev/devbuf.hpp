// device buffers and host-side helpers of the loaders -- part of the one translation unit device.hip (included there, in order; not a header of its own).
// ------------------------------------------------------------------------------------
// Host-side plumbing
// ------------------------------------------------------------------------------------
// PSIGPU_POISON=<byte> (debugging aid): every fresh device allocation is filled with that byte, so that a kernel reading
// what nobody wrote gives the same wrong answer every time instead of whatever the memory held before
static int poison_byte()
{
  static const int b = [] { const char* e = getenv("PSIGPU_POISON"); return e ? (int)(strtoul(e, nullptr, 0) & 0xFF) : -1; }();
  return b;
}

// ---- the ends of the raw engine copies: a pool that lives as long as the process -------------------------------------------
// The host entry moves reads in and records out with hsa_amd_memory_async_copy_on_engine, of which HIP knows nothing.
// THE LIFETIME RULE (round 6; it replaces round 5's "freed a second later"): every buffer that is ever the source or the
// destination of such a copy -- device or pinned host -- is taken from this pool and goes back to it, never straight to
// hipFree / hipHostFree.  The pool hands a returned buffer to the next taker (contexts come and go, a chunk loop regrows:
// the same few buffers go round), and memory leaves it only in trim(), when more than a budget sits idle, and only after
// a QUIESCE that began after the buffer came back: one marker copy on every engine queue the library has used, each
// waited for (a queue is in order: everything submitted before the marker, trailing packets included, has been consumed),
// then hipDeviceSynchronize.  No clock is involved.
struct EndPool {
  struct Buf { void* p; size_t cap; bool host; unsigned flags; int device; };
  std::mutex mu;
  std::vector<Buf> idle, live;
  size_t idle_bytes[2] = { 0, 0 };             // [0] device, [1] pinned host
  size_t budget[2] = { (size_t)8 << 30, (size_t)4 << 30 };
  uint64_t n_alloc = 0, n_reuse = 0, n_trim_freed = 0, n_quiesce = 0;
  void (*quiesce)() = nullptr;                 // set by the host entry once it knows the engines (dev/host_entry.inc)
  // sizes are rounded up to m * 2^e, m in 4..7: a regrown buffer leaves at most a quarter unused and classes are few
  static size_t size_class(size_t bytes)
  {
    if (bytes < 4096) return 4096;
    int e = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t q = (size_t)1 << (e - 2);
    return (bytes + q - 1) & ~(q - 1);
  }
  void* take(bool host, size_t bytes, unsigned flags, int device, size_t* cap_out)
  {
    const size_t want = size_class(bytes);
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = idle.size();
      for (size_t i = 0; i < idle.size(); ++i) {
        const Buf& b = idle[i];
        if (b.host != host || b.flags != flags || (!host && b.device != device) || b.cap < bytes || b.cap > 2 * want) continue;
        if (best == idle.size() || b.cap < idle[best].cap) best = i;
      }
      if (best != idle.size()) {
        Buf b = idle[best];
        idle.erase(idle.begin() + best);
        idle_bytes[host] -= b.cap;
        live.push_back(b);
        ++n_reuse;
        *cap_out = b.cap;
        return b.p;
      }
    }
    void* p = nullptr;
    hipError_t e = host ? hipHostMalloc(&p, want, flags) : hipMalloc(&p, want);
    if (e != hipSuccess) {                     // out of memory with buffers idle: let them go (by the rule) and try once more
      (void)hipGetLastError();
      trim(true);
      e = host ? hipHostMalloc(&p, want, flags) : hipMalloc(&p, want);
      if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    if (!host && poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), want); (void)hipDeviceSynchronize(); }
    std::lock_guard<std::mutex> lk(mu);
    live.push_back(Buf{ p, want, host, flags, device });
    ++n_alloc;
    *cap_out = want;
    return p;
  }
  bool owns(const void* p)
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const Buf& b : live) if (b.p == p) return true;
    return false;
  }
  void give(void* p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < live.size(); ++i)
      if (live[i].p == p) {
#ifdef PSIGPU_CAMPAIGN
        // (campaign builds only, arm C: straight back to HIP as until round 5 -- is THAT what the load campaigns caught?)
        if (getenv("PSIGPU_AB_EARLY_FREE")) {
          if (live[i].host) (void)hipHostFree(p); else (void)hipFree(p);
          live.erase(live.begin() + i);
          return;
        }
#endif
        idle.push_back(live[i]);
        idle_bytes[live[i].host] += live[i].cap;
        live.erase(live.begin() + i);
        return;
      }
    // not ours: the caller handed back something this pool never gave out -- leave it alone (a leak is the lesser evil)
  }
  // more idle than the budget (or everything, on request): the oldest idle buffers leave the process, after a quiesce
  void trim(bool all = false)
  {
    std::vector<Buf> out;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (int h = 0; h < 2; ++h) {
        if (!all && idle_bytes[h] <= budget[h]) continue;
        const size_t keep_below = all ? 0 : budget[h] / 2;
        for (size_t i = 0; i < idle.size() && idle_bytes[h] > keep_below;) {
          if ((int)idle[i].host != h) { ++i; continue; }
          out.push_back(idle[i]);
          idle_bytes[h] -= idle[i].cap;
          idle.erase(idle.begin() + i);
        }
      }
    }
    if (out.empty()) return;
    // (the candidates are out of the idle list: nobody can take them while the queues drain)
    if (quiesce) { quiesce(); ++n_quiesce; }
    (void)hipDeviceSynchronize();
    for (const Buf& b : out) { if (b.host) (void)hipHostFree(b.p); else (void)hipFree(b.p); }
    std::lock_guard<std::mutex> lk(mu);
    n_trim_freed += out.size();
  }
};
static EndPool g_ends;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool pooled = false;            // an end of the host entry's engine copies: memory from g_ends, back to g_ends
  int device = 0;                 // (pooled buffers: the device they live on)
  hipError_t ensure(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (pooled) {
      if (p) { g_ends.give(p); p = nullptr; cap = 0; }
      p = g_ends.take(false, bytes + bytes / 8 + 256, 0, device, &cap);
      if (!p) { cap = 0; return hipErrorOutOfMemory; }
      return hipSuccess;
    }
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) { cap = want; if (poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), want); (void)hipDeviceSynchronize(); } }
    return e;
  }
  void release()
  {
    if (p) { if (pooled) g_ends.give(p); else (void)hipFree(p); }
    p = nullptr; cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// pinned host memory that is an end of an engine copy (staging, landing buffers): same pool, same rule
struct HostEnd {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes, unsigned flags)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) { g_ends.give(p); p = nullptr; cap = 0; }
    p = g_ends.take(true, bytes, flags, 0, &cap);
    if (!p) { cap = 0; return hipErrorOutOfMemory; }
    return hipSuccess;
  }
  void release() { if (p) g_ends.give(p); p = nullptr; cap = 0; }
};

struct TmpBuf {           // scoped device allocation (table construction)
  void* p = nullptr;
  ~TmpBuf() { drop(); }
  void drop() { if (p) (void)hipFree(p); p = nullptr; }
  hipError_t alloc(size_t bytes)
  {
    drop();
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipSuccess && poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), bytes ? bytes : 16); (void)hipDeviceSynchronize(); }
    return e;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// fn(begin, end) over [0, n) on a few host threads (index-load time loops over all nodes / bases)
template <typename F>
void parallel_for(uint64_t n, uint64_t grain, F fn)
{
  unsigned hw = std::thread::hardware_concurrency();
  uint64_t parts = std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 32), (n + grain - 1) / std::max<uint64_t>(1, grain));
  if (parts <= 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  const uint64_t per = (n + parts - 1) / parts;
  for (uint64_t t = 1; t < parts; ++t) th.emplace_back([=] { fn(std::min(n, t * per), std::min(n, (t + 1) * per)); });
  fn(0, std::min(n, per));
  for (auto& t : th) t.join();
}

