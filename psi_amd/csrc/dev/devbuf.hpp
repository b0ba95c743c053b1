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

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) { cap = want; if (poison_byte() >= 0) { (void)hipMemset(p, poison_byte(), want); (void)hipDeviceSynchronize(); } }
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
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

