// host entry: pinned-memory test, staging worker, widening threads -- part of the one translation unit device.hip (included there, in order; not a header of its own).
namespace {

// is this host pointer pinned (hipHostMalloc / hipHostRegister), i.e. can the copy engine read it in place?
// `delta` = what to add to the host address to get the address the device side uses for the same
// byte (0 for hipHostMalloc memory; memory pinned later with hipHostRegister may be mapped elsewhere)
bool host_ptr_is_pinned(const void* p, ptrdiff_t* delta)
{
  *delta = 0;
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeHost) return false;
  if (a.devicePointer && a.hostPointer) *delta = (const char*)a.devicePointer - (const char*)a.hostPointer;
  return true;
}

// pageable -> pinned staging copy with a few threads (one core moves ~10 GB/s, the link 55)
void parallel_copy(char* dst, const char* src, size_t n)
{
  const size_t MIN_PART = 2u << 20;
  unsigned parts = (unsigned)std::min<size_t>(4, std::max<size_t>(1, n / MIN_PART));
  if (parts <= 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n / parts + 63) & ~(size_t)63;
  for (unsigned t = 1; t < parts; ++t) {
    const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per);
    th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
  }
  memcpy(dst, src, std::min(n, per));
  for (auto& t : th) t.join();
}

}  // namespace

namespace {

// Widening of the wire records (k_hits_wire16) into the caller's 32-byte records on a few host threads, sub-batch
// after sub-batch, while the pipeline goes on.  Job j: `n` records from a slot's pinned landing buffer to `dst`;
// thread 0 waits for the slot's transfer, then every thread widens its slice.
// one 32-byte record into the caller's (pinned) array with two streaming stores: the array is written once, front
// to back, 224 MB per 1 M-read chunk -- ordinary stores would first READ every line they fill
static inline void store_hit(psigpu_hit* dst, uint64_t node, uint64_t noff, uint64_t read, uint64_t roff)
{
  typedef long long v2 __attribute__((vector_size(16)));
  v2 lo = { (long long)node, (long long)noff }, hi = { (long long)read, (long long)roff };
  __builtin_nontemporal_store(lo, reinterpret_cast<v2*>(dst));
  __builtin_nontemporal_store(hi, reinterpret_cast<v2*>(dst) + 1);
}

// One helper thread that lives with the context and runs one job at a time (the stager of pageable reads: a call used to
// start and join a thread of its own).
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::function<void()> job;
  bool has = false, busy = false, quit = false;
  void run(std::function<void()> f)
  {
    { std::lock_guard<std::mutex> lk(mu); job = std::move(f); has = true; busy = true; }
    if (!th.joinable()) th = std::thread([this] { loop(); });
    cv.notify_one();
  }
  void wait()                                   // the job, if any, has returned
  {
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return !busy; });
  }
  void loop()
  {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || has; });
        if (quit) return;
        f = std::move(job); has = false;
      }
      f();
      { std::lock_guard<std::mutex> lk(mu); busy = false; }
      cv_done.notify_all();
    }
  }
  ~Worker()
  {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

// The CPUs of the NUMA node the device hangs off (/sys/bus/pci/devices/<bdf>/numa_node, .../node<N>/cpulist): the
// library's own threads -- they read what the copy engine just wrote into pinned memory next to the device and write the
// caller's records -- stay there (round 5: the boxes have two nodes and a thread that lands on the far one does both
// across the socket link).  Nothing is bound when anything about this cannot be read.
struct NodeCpus {
  cpu_set_t set; bool ok = false;
  static NodeCpus of_device(int device)
  {
    NodeCpus nc; CPU_ZERO(&nc.set);
    char bdf[64] = { 0 };
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return nc; }
    for (char* c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    int node = -1;
    if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return nc;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return nc;
    char list[1024] = { 0 };
    const bool got = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!got) return nc;
    cpu_set_t allowed; CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return nc;
    int n = 0;
    for (char* p = list; *p && *p != '\n';) {
      char* e = nullptr;
      long a = strtol(p, &e, 10), b = a;
      if (e == p) break;
      if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); if (e == p) break; }
      for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &nc.set); ++n; }
      p = *e == ',' ? e + 1 : e;
    }
    nc.ok = n >= 2;
    return nc;
  }
  void bind_this_thread() const { if (ok) (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set); }
};

struct Widener {
  struct Job { const void* src; psigpu_hit* dst; uint64_t n, id_base, rec_base; int slot; WireFmt fmt; };
  // The threads live as long as the context (round 4: a call used to start and join up to eight threads of its own,
  // 0.15-0.2 ms of a 2-ms call); a call is a SESSION: begin() resets the job list and wakes them, end() waits until
  // every one of them has left the session.
  std::vector<Job> jobs;
  size_t n_jobs = 0;
  std::atomic<size_t> posted{ 0 }, ready{ 0 };
  // slices done, PER JOB: thread 0 may be a job ahead of a thread that was descheduled inside the job before, so a
  // count over all jobs reaches "T x (j + 1)" while a slice of job j is still being read (seen under three fuzz
  // processes on one box: a record of the sub-batch that reused the landing buffer)
  std::unique_ptr<std::atomic<uint32_t>[]> parts;
  size_t parts_cap = 0;
  size_t checked = 0;                           // caller's thread only: jobs [0, checked) are known to be finished
  std::atomic<bool> stop{ false };              // the session is abandoned (error path): leave it
  std::vector<std::thread> th;
  std::function<void(int)> wait_copy;           // blocks until the slot's device-to-host transfer is complete
  // PSIGPU_TRACE=2 (timeline of the two-in-flight path): when job j's transfer was seen complete / its last slice widened,
  // milliseconds on the caller's clock
  double* tl_copy = nullptr; double* tl_wide = nullptr; double tl_base = 0;
  static double tl_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  unsigned T = 0;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t session = 0;                         // (mu)
  bool quit = false;                            // (mu)
  std::atomic<unsigned> left{ 0 };              // threads that have left the current session
  bool open = false;                            // caller's thread only: a session is running

  void begin(unsigned n_threads, size_t n_jobs_, std::function<void(int)> wc)
  {
    if (T == 0) {
      T = n_threads;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([this, t] { loop(t); });
    }
    n_jobs = n_jobs_;
    if (jobs.size() < n_jobs) jobs.resize(n_jobs);
    if (parts_cap < n_jobs) { parts_cap = n_jobs + n_jobs / 2 + 16; parts.reset(new std::atomic<uint32_t>[parts_cap]); }
    for (size_t j = 0; j < n_jobs; ++j) parts[j].store(0, std::memory_order_relaxed);
    posted.store(0); ready.store(0); stop.store(false); left.store(0);
    checked = 0;
    wait_copy = std::move(wc);
    { std::lock_guard<std::mutex> lk(mu); ++session; }
    cv.notify_all();
    open = true;
  }
  // every thread out of the session (all jobs done, or `stop` after an error): nothing of the call's buffers is touched after this
  void end()
  {
    if (!open) return;
    stop.store(true);
    while (left.load(std::memory_order_acquire) < T) std::this_thread::yield();
    open = false;
  }
  void post(size_t j, const Job& job) { jobs[j] = job; posted.store(j + 1, std::memory_order_release); }
  // every slice of jobs [0, upto) has been widened (called by the thread that posts)
  void wait_finished(size_t upto)
  {
    for (; checked < upto; ++checked)
      while (parts[checked].load(std::memory_order_acquire) < T) std::this_thread::yield();
  }
  // the same question without waiting (the host entry's loop asks it while it has other things to look at)
  bool finished(size_t upto)
  {
    for (; checked < upto; ++checked)
      if (parts[checked].load(std::memory_order_acquire) < T) return false;
    return true;
  }
  NodeCpus near;                                // where the threads run (set before the first begin())
  void loop(unsigned t)
  {
    near.bind_this_thread();
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || session != seen; });
        if (quit) return;
        seen = session;
      }
      run(t);
      left.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void run(unsigned t)
  {
    for (size_t j = 0; j < n_jobs; ++j) {
      while (posted.load(std::memory_order_acquire) <= j) { if (stop.load()) return; std::this_thread::yield(); }
      const Job job = jobs[j];
      if (t == 0) { if (job.n) wait_copy(job.slot); if (tl_copy) tl_copy[j] = tl_now() - tl_base; ready.store(j + 1, std::memory_order_release); }
      else while (ready.load(std::memory_order_acquire) <= j) { if (stop.load()) return; std::this_thread::yield(); }
      const uint64_t a = job.n * t / T, b = job.n * (t + 1) / T;
      if (job.fmt.bytes && job.fmt.bytes < 8) {
        // packed records (k_hits_wirep): blocks of WP_BLOCK, each behind the read id of its first record
        const char* src = static_cast<const char*>(job.src);
        const uint32_t B = job.fmt.bytes, nb = job.fmt.noff_bits, vb = job.fmt.node_bits, rb = job.fmt.roff_bits;
        const uint64_t nm = (1ull << nb) - 1, vm = (1ull << vb) - 1, rm = (1ull << rb) - 1, km = (1ull << (8 * B)) - 1;
        const uint64_t stride = wirep_block_stride(B), div = job.fmt.roff_div;
        for (uint64_t i = a; i < b;) {
          const uint64_t blk = i / WP_BLOCK, e = std::min<uint64_t>(b, (blk + 1) * WP_BLOCK);
          const char* p = src + blk * stride;
          uint64_t first; memcpy(&first, p, 8);
          const uint64_t rid0 = job.rec_base + (first & 0xFFFFFFFFull);
          p += 8 + (i % WP_BLOCK) * B;
          for (; i < e; ++i, p += B) {
            uint64_t key; memcpy(&key, p, 8);         // (the landing buffer has room behind the last record)
            key &= km;
            const uint64_t noff = key & nm; key >>= nb;
            const uint64_t node = key & vm; key >>= vb;
            const uint64_t roff = (key & rm) * div; key >>= rb;
            store_hit(job.dst + i, job.id_base + node, noff, rid0 + key, roff);
          }
        }
      } else if (job.fmt.bytes == 8) {
        const uint64_t* src = static_cast<const uint64_t*>(job.src);
        const uint32_t nb = job.fmt.noff_bits, vb = job.fmt.node_bits, rb = job.fmt.roff_bits;
        const uint64_t nm = (1ull << nb) - 1, vm = (1ull << vb) - 1, rm = (1ull << rb) - 1;
        for (uint64_t i = a; i < b; ++i) {
          uint64_t key = src[i];
          const uint64_t noff = key & nm; key >>= nb;
          const uint64_t node = key & vm; key >>= vb;
          const uint64_t roff = key & rm; key >>= rb;
          store_hit(job.dst + i, job.id_base + node, noff, job.rec_base + key, roff);
        }
      } else {
        const uint4* src = static_cast<const uint4*>(job.src);
        for (uint64_t i = a; i < b; ++i) {
          const uint4 w = src[i];
          store_hit(job.dst + i, job.id_base + w.x, w.y, job.rec_base + w.z, w.w);
        }
      }
      __builtin_ia32_sfence();                   // the streaming stores above, before the slice is declared done
      if (parts[j].fetch_add(1, std::memory_order_acq_rel) + 1 == T && tl_wide) tl_wide[j] = tl_now() - tl_base;
    }
  }
  ~Widener()
  {
    end();
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    for (auto& x : th) if (x.joinable()) x.join();
  }
};

}  // namespace

