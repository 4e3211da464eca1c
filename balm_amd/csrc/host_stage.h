// Host -> HBM uploads of the C ABI's big caller-owned arrays (cluster tables, point clouds, scan ids).
//
// The caller's memory is pageable; hipMemcpyAsync from it is the runtime's own bounce copy, one thread, 10-14 GB/s of a
// 63 GB/s link (round 4: 161 MB of points in 15 ms in front of a 3.6 ms association).  Here a ring of pinned chunks is
// filled by a small pool of host threads (15 + the caller on a big host) -- memcpy of the caller's array, or the caller's fill callback writing its
// clusters straight into the chunk (balm_set_features_cb: no flattened copy in between) -- while the DMA engine drains the
// previous chunk.  One wake-up of the pool per upload, not per chunk: the workers take (chunk, slice) tasks off one
// counter and wait for the chunk's buffer to be free; the calling thread only issues the DMAs and frees buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace balm {

// The CPUs next to a GPU (sysfs: local_cpulist of its PCI function).  On a two-socket host the DMA engine reads a pinned chunk that
// lives on the OTHER socket at 21-23 GB/s instead of 55 (profiles/r05f_stage_trace.txt: fills at 110 GB/s, DMAs at 21), and the
// kernel places a chunk where the allocating thread runs: the ring is therefore allocated and filled by pool threads that sit on the
// GPU's own node -- the library pins ITS threads, never the caller's.  (A process pinned to that node by its launcher: 53 GB/s with or
// without this; unpinned: 34 -> 5x GB/s with it.)
struct GpuNode { bool valid = false; cpu_set_t cpus; };
inline GpuNode gpu_local_cpus(int device) {
  GpuNode g;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return g;
  for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE *f = fopen(path, "r");
  if (!f) return g;
  char line[1024] = {0};
  const bool ok = fgets(line, sizeof(line), f) != nullptr;
  fclose(f);
  if (!ok) return g;
  CPU_ZERO(&g.cpus);
  int n = 0;
  for (char *p = line; *p && *p != '\n';) {          // "0-63,128-191"
    char *e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &g.cpus); n++; }
    if (*p == ',') p++;
  }
  g.valid = n > 0;
  return g;
}

// The NUMA node a caller's buffer lives on (move_pages with no target nodes only reports), and that node's CPUs.  A strided gather
// reads four bytes of the caller's clouds for every byte it writes into the pinned chunk: with the clouds on the OTHER socket the
// pool's reads cross the socket link at 105-115 GB/s (5.6 ms for the shipped window instead of 3.55, profiles/r06_gather_numa.txt), so
// for those uploads the pool goes where the SOURCE is and only the packed quarter crosses.
inline int numa_node_of(const void *p) {
#if defined(__linux__) && defined(SYS_move_pages)
  void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095);
  int status = -1;
  if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) == 0 && status >= 0) return status;
#endif
  return -1;
}
inline GpuNode numa_node_cpus(int node) {
  GpuNode g;
  if (node < 0) return g;
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return g;
  char line[1024] = {0};
  const bool ok = fgets(line, sizeof(line), f) != nullptr;
  fclose(f);
  if (!ok) return g;
  CPU_ZERO(&g.cpus);
  int n = 0;
  for (char *q = line; *q && *q != '\n';) {
    char *e = nullptr;
    const long a = strtol(q, &e, 10);
    if (e == q) break;
    long b = a;
    q = e;
    if (*q == '-') { b = strtol(q + 1, &e, 10); q = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &g.cpus); n++; }
    if (*q == ',') q++;
  }
  g.valid = n > 0;
  return g;
}

// The cores of `set` (a core = the CPUs of one thread_siblings_list, read from sysfs) cut into n contiguous groups: group q for pool thread q.
// Fewer cores than threads: every group is the whole set.  (HostPool::spread_cpus says why; tests/cpp/host_stage_test.cpp checks the partition.)
inline std::vector<cpu_set_t> core_groups(const cpu_set_t &set, int n) {
  std::vector<cpu_set_t> cores;                  // one entry per core: its hardware threads inside `set`
  for (int c = 0; c < CPU_SETSIZE; c++) {
    if (!CPU_ISSET(c, &set)) continue;
    char path[128], line[128] = {0};
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    cpu_set_t sib;
    CPU_ZERO(&sib);
    CPU_SET(c, &sib);
    int lead = c;
    if (FILE *f = fopen(path, "r")) {
      if (fgets(line, sizeof(line), f)) {
        lead = -1;
        for (char *p = line; *p && *p != '\n';) {          // "64,192" or "64-65"
          char *e = nullptr;
          const long x = strtol(p, &e, 10);
          if (e == p) break;
          long y = x;
          p = e;
          if (*p == '-') { y = strtol(p + 1, &e, 10); p = e; }
          for (long q = x; q <= y && q < CPU_SETSIZE; q++) if (CPU_ISSET((int)q, &set)) { if (lead < 0) lead = (int)q; CPU_SET((int)q, &sib); }      // (lead: the core's first CPU inside the set)
          if (*p == ',') p++;
        }
      }
      fclose(f);
    }
    if (lead == c || lead < 0) cores.push_back(sib);
  }
  std::vector<cpu_set_t> per((size_t)std::max(n, 1), set);
  const long nc = (long)cores.size();
  if (nc >= n && n > 0)
    for (int q = 0; q < n; q++) {
      CPU_ZERO(&per[(size_t)q]);
      for (long k = (long)q * nc / n; k < (long)(q + 1) * nc / n; k++) CPU_OR(&per[(size_t)q], &per[(size_t)q], &cores[(size_t)k]);
    }
  return per;
}

#ifndef BALM_POOL_SPREAD
#define BALM_POOL_SPREAD 1          // (tools/ubench_gather.hip: 0 = every pool thread floats over the whole node)
#endif
class HostPool {
 public:
  // Pool 0 is the process's pool.  The device threads of a one-process multi-GPU context (balm_create_multi) each get a pool of
  // their own (key = shard index, 1 ..): their uploads run side by side, each beside ITS GPU, instead of queueing behind one another
  // for the one pool (8 x 400 MB one shard at a time at BASELINE configs[3], VERDICT r5 Weak 3a).
  static constexpr int MAX_POOLS = 64;
  static HostPool &get(int key = 0) {
    static std::mutex mu;
    static std::unique_ptr<HostPool> pools[MAX_POOLS];
    if (key < 0 || key >= MAX_POOLS) key = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!pools[key]) pools[key].reset(new HostPool(key == 0 ? 0 : shard_pool_threads()));      // (the process's pool keeps its full size)
    return *pools[key];
  }
  // Threads of the pools created from now on (0 = by the host's size): balm_create_multi sets it by its device count -- sixteen
  // fillers per device saturate one link (host_stage.h's ring geometry), eight devices' worth of them next to four GPUs' memory
  // controllers do not fit a socket: 8 per pool from five devices on.
  static int &shard_pool_threads() { static int n = 0; return n; }
  int workers() const { return (int)th_.size(); }
  // every pool thread (and the caller) runs fn(thread index) once; returns when all have returned.  One job at a time.
  // aff (optional): the pool threads move to these CPUs first (and stay there until a job names other ones)
  void run_all(const std::function<void(int)> &fn, const GpuNode *aff = nullptr) {
    std::lock_guard<std::mutex> job(job_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; aff_ = (aff && aff->valid) ? aff : nullptr; pending_ = (int)th_.size(); gen_++;
    }
    cv_go_.notify_all();
    fn((int)th_.size());
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  explicit HostPool(int want = 0) {
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc >= 64 ? 15 : hc >= 16 ? 7 : hc >= 4 ? 3 : 1;      // + the calling thread
    if (want > 0 && want - 1 < n) n = want - 1;
#ifdef BALM_HOST_POOL_THREADS                                     // (tools/ubench_h2d.hip: the pipeline by thread count)
    n = BALM_HOST_POOL_THREADS - 1;
#endif
    for (int t = 0; t < n; t++) th_.emplace_back([this, t] { loop(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_go_.notify_all();
    for (auto &t : th_) t.join();
  }
  friend struct std::default_delete<HostPool>;
  void loop(int t) {
    unsigned long seen = 0;
    cpu_set_t cur_aff;
    bool have_aff = false;
    CPU_ZERO(&cur_aff);
    for (;;) {
      const std::function<void(int)> *f;
      const GpuNode *aff;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_go_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_; f = fn_; aff = aff_;
      }
      if (aff && (!have_aff || !CPU_EQUAL(&aff->cpus, &cur_aff))) {
#if BALM_POOL_SPREAD
        const cpu_set_t mine = spread_cpus(aff->cpus, t);      // this thread's own group of the node's cores
        if (pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &mine) == 0) { cur_aff = aff->cpus; have_aff = true; }
#else
        if (pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &aff->cpus) == 0) { cur_aff = aff->cpus; have_aff = true; }
#endif
      }
      (*f)(t);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) cv_done_.notify_all();
      }
    }
  }
  // The CPUs of pool thread t: the cores of `set` cut into one contiguous group per thread (a core = the CPUs of a thread_siblings_list), so that
  // two fillers never share a core's hardware threads, the sixteen are spread over the socket's dies, and the scheduler still has a few cores to
  // choose from when another tenant's thread sits on one (ONE core per thread: 6-10 ms uploads whenever that happens, measured).  Floating over the
  // whole node instead, a process keeps the placement its threads started with -- uploads of 3.2, 3.6, 4.4, 4.7 ms from one process to the next;
  // grouped: 3.1-3.3 (tools/ubench_gather.hip, profiles/r06_cpp_leg_numa.txt).  Read from sysfs once per CPU set.
  std::vector<std::pair<cpu_set_t, std::vector<cpu_set_t>>> groups_;
  std::mutex groups_mu_;
  cpu_set_t spread_cpus(const cpu_set_t &set, int t) {
    const int n = (int)th_.size();
    std::lock_guard<std::mutex> lk(groups_mu_);
    for (auto &g : groups_) if (CPU_EQUAL(&g.first, &set)) return g.second[(size_t)t];
    std::vector<cpu_set_t> per = core_groups(set, n);
    groups_.emplace_back(set, per);
    return groups_.back().second[(size_t)t];
  }
  std::vector<std::thread> th_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_go_, cv_done_;
  const std::function<void(int)> *fn_ = nullptr;
  const GpuNode *aff_ = nullptr;
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool quit_ = false;
};

// memcpy with streaming stores into a pinned chunk.  glibc's memcpy writes a 256 KiB slice through the caches: every destination line
// is first READ (write allocate), the pool's bursts then move three bytes for every byte copied -- 16 threads at 110 GB/s are 330 GB/s
// of a socket's ~460 -- and the DMA engine reading the previous chunk beside them fell to 28 GB/s (profiles/r05h_stage_trace.txt).
// Streaming stores skip the read and leave the caches to the caller.  dst must be 16-byte aligned (chunks and slices are).
#if defined(__x86_64__)
#include <emmintrin.h>
inline void stream_copy(void *dst, const void *src, size_t n) {
  char *d = static_cast<char *>(dst);
  const char *s = static_cast<const char *>(src);
  if ((reinterpret_cast<uintptr_t>(d) & 15) != 0 || n < 64) { std::memcpy(d, s, n); return; }
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i));
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 16));
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 32));
    const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 48));
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i), a);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 16), b);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 32), c);
    _mm_stream_si128(reinterpret_cast<__m128i *>(d + i + 48), e);
  }
  if (i < n) std::memcpy(d + i, s + i, n - i);
  _mm_sfence();
}
#else
inline void stream_copy(void *dst, const void *src, size_t n) { std::memcpy(dst, src, n); }
#endif

// Point containers the caller already holds (balm_associate_scans, balm_build_clusters_planes, balm_window_add_scan_strided): `n`
// arrays of `count[k]` elements `stride` bytes apart with float x, y, z at offset 0 of each element -- pcl::PointCloud<PointXYZINormal>
// is 48-byte elements (the reference's PointType, include/tools.hpp:22).  The pool threads read the clouds where they lie and write
// PACKED records into the pinned chunk: 12-byte xyz, or (aux_off != NO_AUX) 16-byte xyz + the float at byte `aux_off` of the element
// (the observing pose the virtual benchmark keeps in `intensity`, benchmark_virtual.cpp:586).  Nothing is flattened on the caller's
// thread and the per-point scan index never exists on the host: the device expands it from the counts.
struct StridedPoints {
  static constexpr size_t NO_AUX = ~(size_t)0;
  int n = 0;
  const void *const *base = nullptr;
  size_t stride = 0, aux_off = NO_AUX;
  std::vector<long> first;      // first[k] = points before array k; first[n] = all points
  size_t rec() const { return aux_off == NO_AUX ? 12 : 16; }
  long total() const { return first.empty() ? 0 : first.back(); }
  bool set(int n_, const void *const *base_, const long *count, size_t stride_, size_t aux_off_ = NO_AUX) {
    n = n_; base = base_; stride = stride_; aux_off = aux_off_;
    if (n < 0 || (n > 0 && (!base || !count)) || stride < 12 || (stride & 3) != 0) return false;
    if (aux_off != NO_AUX && (aux_off + 4 > stride || (aux_off & 3) != 0)) return false;
    first.assign((size_t)n + 1, 0);
    for (int k = 0; k < n; k++) {
      if (count[k] < 0 || (count[k] > 0 && !base[k])) return false;
      first[(size_t)k + 1] = first[(size_t)k] + count[k];
    }
    return true;
  }
  // packed records of points [p0, p0 + np) of the concatenation -> dst
  void gather(char *dst, long p0, long np) const {
    if (np <= 0) return;
    int k = (int)(std::upper_bound(first.begin(), first.end(), p0) - first.begin()) - 1;
    while (np > 0) {
      while (first[(size_t)k + 1] <= p0) k++;
      const long take = std::min(np, first[(size_t)k + 1] - p0);
      const char *src = static_cast<const char *>(base[k]) + (size_t)(p0 - first[(size_t)k]) * stride;
      if (aux_off == NO_AUX) gather_xyz(dst, src, take, stride); else gather_xyzw(dst, src, take, stride, aux_off);
      dst += (size_t)take * rec(); p0 += take; np -= take;
    }
  }
  static void gather_xyz(char *dst, const char *src, long np, size_t stride) {
    if (stride == 12) { stream_copy_any(dst, src, (size_t)np * 12); return; }
    float *o = reinterpret_cast<float *>(dst);
    long i = 0;
#if defined(__x86_64__)
    // a 16-byte load per element stays inside it (stride >= 16 here); four points leave as three aligned streaming stores:
    // [x0 y0 z0 x1] [y1 z1 x2 y2] [z2 x3 y3 z3].  The output advances 12 bytes per point: at most three scalar points re-align it.
    while (i < np && (reinterpret_cast<uintptr_t>(o) & 15) != 0) {
      const float *p = reinterpret_cast<const float *>(src + (size_t)i * stride);
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o += 3; i++;
    }
    for (; i + 4 <= np; i += 4) {
      const char *s = src + (size_t)i * stride;
#ifdef BALM_GATHER_PREFETCH                      // (tools/ubench_gather.hip: bytes ahead)
      _mm_prefetch(s + BALM_GATHER_PREFETCH, _MM_HINT_NTA); _mm_prefetch(s + BALM_GATHER_PREFETCH + 64, _MM_HINT_NTA);
      _mm_prefetch(s + BALM_GATHER_PREFETCH + 128, _MM_HINT_NTA);
#endif
      const __m128 a = _mm_loadu_ps(reinterpret_cast<const float *>(s));
      const __m128 b = _mm_loadu_ps(reinterpret_cast<const float *>(s + stride));
      const __m128 c = _mm_loadu_ps(reinterpret_cast<const float *>(s + 2 * stride));
      const __m128 d = _mm_loadu_ps(reinterpret_cast<const float *>(s + 3 * stride));
      // r0 = a0 a1 a2 b0
      const __m128 ab = _mm_shuffle_ps(a, b, _MM_SHUFFLE(0, 0, 2, 2));        // a2 a2 b0 b0
      const __m128 r0 = _mm_shuffle_ps(a, ab, _MM_SHUFFLE(2, 0, 1, 0));       // a0 a1 a2 b0
      // r1 = b1 b2 c0 c1
      const __m128 r1 = _mm_shuffle_ps(b, c, _MM_SHUFFLE(1, 0, 2, 1));        // b1 b2 c0 c1
      // r2 = c2 d0 d1 d2
      const __m128 cd = _mm_shuffle_ps(c, d, _MM_SHUFFLE(0, 0, 2, 2));        // c2 c2 d0 d0
      const __m128 r2 = _mm_shuffle_ps(cd, d, _MM_SHUFFLE(2, 1, 2, 0));       // c2 d0 d1 d2
      _mm_stream_ps(o, r0); _mm_stream_ps(o + 4, r1); _mm_stream_ps(o + 8, r2);
      o += 12;
    }
#endif
    for (; i < np; i++) {
      const float *p = reinterpret_cast<const float *>(src + (size_t)i * stride);
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o += 3;
    }
#if defined(__x86_64__)
    _mm_sfence();
#endif
  }
  static void gather_xyzw(char *dst, const char *src, long np, size_t stride, size_t aux) {
    float *o = reinterpret_cast<float *>(dst);
    for (long i = 0; i < np; i++) {
      const float *p = reinterpret_cast<const float *>(src + (size_t)i * stride);
      const float w = *reinterpret_cast<const float *>(src + (size_t)i * stride + aux);
#if defined(__x86_64__)
      if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) { _mm_stream_ps(o, _mm_set_ps(w, p[2], p[1], p[0])); o += 4; continue; }
#endif
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = w; o += 4;
    }
#if defined(__x86_64__)
    _mm_sfence();
#endif
  }
  static void stream_copy_any(void *dst, const void *src, size_t n) {      // stream_copy wants a 16-byte aligned destination
    char *d = static_cast<char *>(dst);
    const char *s = static_cast<const char *>(src);
    const size_t head = std::min(n, (size_t)((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15));
    std::memcpy(d, s, head);
    stream_copy(d + head, s + head, n - head);
  }
};

// fn(lo, hi) over [0, n) in contiguous pieces, one or a few per pool thread; serial below `grain`
inline void parallel_ranges(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn, int pool_key = 0) {
  if (n == 0) return;
  HostPool &pool = HostPool::get(pool_key);
  if (n <= grain || pool.workers() == 0) { fn(0, n); return; }
  const size_t pieces = std::min<size_t>((n + grain - 1) / grain, (size_t)(pool.workers() + 1) * 4);
  std::atomic<size_t> next{0};
  pool.run_all([&](int) {
    for (;;) {
      const size_t q = next.fetch_add(1, std::memory_order_relaxed);
      if (q >= pieces) return;
      fn(n * q / pieces, n * (q + 1) / pieces);
    }
  });
}

// How a filled chunk crosses the link: a KERNEL that pulls it out of the pinned buffer (256 workgroups: 56-57 GB/s, the same as the
// DMA engines at their best, tools/ubench_h2d.hip) instead of hipMemcpyAsync.  The runtime's copy path chooses its engine by rules
// of its own: the first ~50 copies of a process ran at 27 GB/s, every later one at 52 (profiles/r05j_upload_cadence.txt: seven
// balm_associate calls at 7.6 ms of upload, then 4.1 ms for good -- whatever the idle time or the arrays in between).  The uploads of
// this library come in front of device work that cannot start without them, so the CUs the pull occupies are idle anyway.
#ifdef __HIPCC__
static __global__ __launch_bounds__(256) void k_stage_pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                          const char *__restrict__ tail_src, char *__restrict__ tail_dst, int tail) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}
inline hipError_t stage_enqueue(void *d_dst, const char *pinned, size_t len, hipStream_t stream) {
  const size_t n16 = len / 16;
  const int tail = (int)(len - n16 * 16);
  if ((reinterpret_cast<uintptr_t>(d_dst) & 15) != 0) return hipMemcpyAsync(d_dst, pinned, len, hipMemcpyHostToDevice, stream);
  hipLaunchKernelGGL(k_stage_pull, dim3(256), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(pinned), reinterpret_cast<uint4 *>(d_dst), n16,
                     pinned + n16 * 16, static_cast<char *>(d_dst) + n16 * 16, tail);
  return hipGetLastError();
}
#else
inline hipError_t stage_enqueue(void *d_dst, const char *pinned, size_t len, hipStream_t stream) {
  return hipMemcpyAsync(d_dst, pinned, len, hipMemcpyHostToDevice, stream);
}
#endif

struct PinnedRing {
// Ring geometry, measured on the box (tools/ubench_h2d.hip, profiles/r05c_ubench_h2d.txt; 512 MB from a fresh pageable buffer; the link
// does 55-57 GB/s from pinned memory): 16 threads, 3 x 32 MB: 51-54 GB/s; 3 x 16 MB: 43-48; 4 x 8 MB: 34; 4 x 4 MB: 40-44; 8 threads: 33;
// 32 threads: 35 (the spinning workers get in each other's way).  The runtime's own pageable path: 27-54 GB/s on a fresh buffer.
#ifndef BALM_STAGE_NBUF
#define BALM_STAGE_NBUF 3
#endif
#ifndef BALM_STAGE_CHUNK_MB
#define BALM_STAGE_CHUNK_MB 32
#endif
  static constexpr int NBUF = BALM_STAGE_NBUF;
  static constexpr size_t CHUNK = (size_t)BALM_STAGE_CHUNK_MB << 20;
  char *buf[NBUF] = {};
  hipEvent_t ev[NBUF] = {};
  bool busy[NBUF] = {};      // a DMA out of this buffer was enqueued and its event not yet waited for

  int pool_key = 0;            // whose host threads fill this ring (HostPool::get)
  GpuNode node;                // the CPUs next to the device the ring feeds (invalid: unknown, nobody is pinned)
  int pos = 0;                 // buffer of the next upload's first chunk: an upload continues round the ring where the previous one stopped,
                               // so its first chunks fill buffers that are already free while the previous upload's last DMAs still run

  // allocated (and first touched) by a pool thread that sits on the device's own node
  hipError_t init(int device) {
    if (buf[0]) return hipSuccess;
#ifdef BALM_COLD_TRACE
    const auto tr0 = std::chrono::steady_clock::now();
    struct Done { std::chrono::steady_clock::time_point t; ~Done() { fprintf(stderr, "[cold] PinnedRing::init %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } done{tr0};
#endif
    node = gpu_local_cpus(device);
    hipError_t err = hipSuccess;
    auto alloc = [&]() {
      if (hipSetDevice(device) != hipSuccess) { err = hipErrorInvalidValue; return; }
      for (int b = 0; b < NBUF; b++) {
        hipError_t e = hipHostMalloc((void **)&buf[b], CHUNK, hipHostMallocDefault);
        if (e == hipSuccess) { std::memset(buf[b], 0, CHUNK); e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming); }
        if (e != hipSuccess) { err = e; return; }
      }
    };
    HostPool &pool = HostPool::get(pool_key);
    if (pool.workers() > 0) pool.run_all([&](int t) { if (t == 0) alloc(); }, &node);
    else alloc();
    if (err != hipSuccess) release();
    return err;
  }
  void release() {
    for (int b = 0; b < NBUF; b++) {
      if (ev[b]) { if (busy[b]) (void)hipEventSynchronize(ev[b]); (void)hipEventDestroy(ev[b]); ev[b] = nullptr; }
      if (buf[b]) { (void)hipHostFree(buf[b]); buf[b] = nullptr; }
      busy[b] = false;
    }
  }
};

// fill(dst, off, len): produce bytes [off, off + len) of the source at dst (pinned).  `unit`: chunk and slice boundaries are
// multiples of it (a cluster-table row, a point).  The copy is ordered on `stream` like a hipMemcpyAsync; on return the
// caller's memory has been read completely (the DMAs out of the ring may still be in flight).
inline hipError_t staged_upload(PinnedRing &ring, int device, hipStream_t stream, void *d_dst, size_t bytes, size_t unit,
                                const std::function<void(char *, size_t, size_t)> &fill, const GpuNode *fill_cpus = nullptr) {
  if (bytes == 0) return hipSuccess;
  hipError_t e = ring.init(device);
  if (e != hipSuccess) return e;
  if (unit == 0 || unit > PinnedRing::CHUNK) return hipErrorInvalidValue;
  if (bytes <= ((size_t)1 << 20)) {      // small: the calling thread alone, no pool wake-up
    const int b = ring.pos;
    if (ring.busy[b]) { (void)hipEventSynchronize(ring.ev[b]); ring.busy[b] = false; }
    fill(ring.buf[b], 0, bytes);
    e = stage_enqueue(d_dst, ring.buf[b], bytes, stream);
    if (e == hipSuccess) e = hipEventRecord(ring.ev[b], stream);
    if (e == hipSuccess) { ring.busy[b] = true; ring.pos = (b + 1) % PinnedRing::NBUF; }
    return e;
  }
  HostPool &pool = HostPool::get(ring.pool_key);
  const long p0 = ring.pos;
  // Chunk plan: the first chunks are small -- 4, 8, 16 MB, then full 32 MB chunks -- so that the link starts after 4 MB have been
  // filled instead of 32 (0.4 of the 3.5 ms the shipped window's points take; the link then never waits: each DMA covers the fill of
  // the next, twice as large chunk).  Chunk k uses buffer (p0 + k) mod NBUF from its start.
#ifndef BALM_STAGE_RAMP_MB
#define BALM_STAGE_RAMP_MB 4
#endif
  const size_t full = PinnedRing::CHUNK / unit * unit;
  std::vector<size_t> cbeg;                      // cbeg[k] = first byte of chunk k; cbeg[nchunks] = bytes
  {
    size_t at = 0, len = std::max(unit, std::min(full, ((size_t)BALM_STAGE_RAMP_MB << 20) / unit * unit));
    if (bytes <= 2 * full) len = full;          // (one or two chunks: nothing to ramp)
    while (at < bytes) { cbeg.push_back(at); at += std::min(len, bytes - at); len = std::min(full, 2 * len / unit * unit); }
    cbeg.push_back(bytes);
  }
  const long nchunks = (long)cbeg.size() - 1;
  // slices: about 256 KiB each so that the threads finish a chunk together, a multiple of the unit
  size_t slice = ((size_t)256 << 10) / unit * unit;
  if (slice == 0) slice = unit;
  std::vector<long> tbeg((size_t)nchunks + 1, 0);      // tbeg[k] = first (chunk, slice) task of chunk k
  for (long k = 0; k < nchunks; k++) tbeg[(size_t)k + 1] = tbeg[(size_t)k] + (long)((cbeg[(size_t)k + 1] - cbeg[(size_t)k] + slice - 1) / slice);
  const long ntasks = tbeg[(size_t)nchunks];
  auto chunk_of = [&](long t) { return (long)(std::upper_bound(tbeg.begin(), tbeg.end(), t) - tbeg.begin()) - 1; };
  std::vector<std::atomic<int>> done((size_t)nchunks);
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  std::atomic<long> next{0}, allowed{-1};      // chunks [0, allowed] may be filled: their buffers are free
  std::atomic<bool> abort{false};
  auto slices_of = [&](long k) { return tbeg[(size_t)k + 1] - tbeg[(size_t)k]; };
  auto run_task = [&](long t, long k) {
    const size_t s = (size_t)(t - tbeg[(size_t)k]);
    const size_t clen = cbeg[(size_t)k + 1] - cbeg[(size_t)k];
    fill(ring.buf[(p0 + k) % PinnedRing::NBUF] + s * slice, cbeg[(size_t)k] + s * slice, std::min(slice, clen - s * slice));
    done[(size_t)k].fetch_add(1, std::memory_order_release);
  };
  hipError_t err = hipSuccess;
#ifdef BALM_STAGE_TRACE          // A/B builds only (tools/build_ab.sh): where a chunk's time goes, on stderr
  std::vector<double> tr_ready((size_t)nchunks), tr_issued((size_t)nchunks), tr_freed((size_t)nchunks);
  const auto tr0 = std::chrono::steady_clock::now();
  auto tr_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(); };
#endif
#ifndef BALM_STAGE_FILL_THREADS
#define BALM_STAGE_FILL_THREADS 16          // pool threads (the caller included) that fill; the others go back to sleep at once
#endif
  auto worker = [&](int tid_) {
    if (tid_ >= BALM_STAGE_FILL_THREADS - 1) return;
    for (;;) {
      const long t = next.fetch_add(1, std::memory_order_relaxed);
      if (t >= ntasks) return;
      const long k = chunk_of(t);
      while (allowed.load(std::memory_order_acquire) < k) {
        if (abort.load(std::memory_order_relaxed)) return;
        std::this_thread::yield();
      }
      run_task(t, k);
    }
  };
  auto driver = [&](int tid) {
    if (tid != pool.workers()) { worker(tid); return; }
    // the calling thread: DMA of chunk k as soon as its slices are in, buffer of chunk k - 1 freed behind it; fills
    // slices itself while it has nothing to issue.  A buffer may still feed a DMA of the PREVIOUS upload through this ring
    // (balm_associate: the scan indices right behind the points): chunk c is granted once the event of its buffer has fired --
    // the first chunk at once, the next ones while the pool fills it, so two uploads in a row leave no bubble on the link.
    long granted = -1;
    auto grant = [&](long upto) -> bool {
      for (long c = granted + 1; c <= upto && c < nchunks; c++) {
        const int gb = (int)((p0 + c) % PinnedRing::NBUF);
        if (ring.busy[gb]) {
          if (hipEventSynchronize(ring.ev[gb]) != hipSuccess) return false;
          ring.busy[gb] = false;
        }
        allowed.store(c, std::memory_order_release);
        granted = c;
      }
      return true;
    };
    if (!grant(PinnedRing::NBUF - 1)) { err = hipErrorInvalidValue; abort.store(true); allowed.store(nchunks, std::memory_order_release); return; }
    for (long k = 0; k < nchunks; k++) {
      const int want = (int)slices_of(k);
      while (done[(size_t)k].load(std::memory_order_acquire) < want) {
        const long t = next.load(std::memory_order_relaxed);
        const long kk = t < ntasks ? chunk_of(t) : nchunks;
        if (kk < nchunks && kk <= allowed.load(std::memory_order_relaxed)) {
          long mine = t;
          if (next.compare_exchange_strong(mine, t + 1, std::memory_order_relaxed)) run_task(t, kk);
        } else {
          std::this_thread::yield();
        }
      }
      const int b = (int)((p0 + k) % PinnedRing::NBUF);
      const size_t len = cbeg[(size_t)k + 1] - cbeg[(size_t)k];
#ifdef BALM_STAGE_TRACE
      tr_ready[(size_t)k] = tr_now();
#endif
      hipError_t e2 = stage_enqueue((char *)d_dst + cbeg[(size_t)k], ring.buf[b], len, stream);
      if (e2 == hipSuccess) e2 = hipEventRecord(ring.ev[b], stream);
      if (e2 != hipSuccess) { err = e2; abort.store(true); allowed.store(nchunks, std::memory_order_release); return; }
      ring.busy[b] = true;
#ifdef BALM_STAGE_TRACE
      tr_issued[(size_t)k] = tr_now();
#endif
      if (k >= 1) {                       // chunk k - 1 + NBUF reuses the buffer of chunk k - 1: wait for that DMA, with DMA k queued behind it
        if (!grant(k - 1 + PinnedRing::NBUF)) { err = hipErrorInvalidValue; abort.store(true); allowed.store(nchunks, std::memory_order_release); return; }
#ifdef BALM_STAGE_TRACE
        tr_freed[(size_t)k - 1] = tr_now();
#endif
      }
    }
  };
  pool.run_all(driver, (fill_cpus && fill_cpus->valid) ? fill_cpus : &ring.node);      // (default: beside the ring, i.e. beside the GPU)
  ring.pos = (int)((p0 + nchunks) % PinnedRing::NBUF);
#ifdef BALM_STAGE_TRACE
  fprintf(stderr, "[stage] %zu bytes, %ld chunks of up to %zu MB, %d threads; per chunk: filled / DMA issued / previous chunk's DMA seen done (ms)\n", bytes, nchunks,
          full >> 20, pool.workers() + 1);
  for (long k = 0; k < nchunks; k++) fprintf(stderr, "[stage]   %3ld  %8.3f %8.3f %8.3f\n", k, tr_ready[(size_t)k], tr_issued[(size_t)k], k + 1 < nchunks ? tr_freed[(size_t)k] : 0.0);
#endif
  return err;
}

// the plain case: a contiguous caller array
inline hipError_t staged_copy(PinnedRing &ring, int device, hipStream_t stream, void *d_dst, const void *src, size_t bytes, size_t unit = 64) {
  if (bytes < ((size_t)1 << 20))      // small: the runtime's own path (one bounce, no pool wake-up)
    return hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream);
  const char *s = static_cast<const char *>(src);
  return staged_upload(ring, device, stream, d_dst, bytes, unit, [s](char *dst, size_t off, size_t len) { stream_copy(dst, s + off, len); });
}

// strided point containers -> packed records on the device (see StridedPoints).  Chunk and slice boundaries fall on multiples of 48
// bytes = four 12-byte records = three 16-byte stores, so that every slice starts on a streaming-store boundary.
#ifndef BALM_GATHER_FOLLOW_SOURCE
#define BALM_GATHER_FOLLOW_SOURCE 1          // (tools/ubench_gather.hip: 0 = the pool stays beside the ring whatever the source)
#endif
inline hipError_t staged_points(PinnedRing &ring, int device, hipStream_t stream, void *d_dst, const StridedPoints &sp) {
  const size_t rec = sp.rec(), bytes = (size_t)sp.total() * rec;
  // where do the caller's containers live?  (a handful of them asked: they were filled by one reader thread)
  GpuNode src_cpus;
  if (BALM_GATHER_FOLLOW_SOURCE && sp.stride >= 2 * rec && bytes > ((size_t)4 << 20)) {
    static std::mutex mu;
    static std::vector<std::pair<int, GpuNode>> known;      // node -> its CPUs, read from sysfs once
    int votes[8] = {0}, asked = 0;
    for (int t = 0; t < 9 && sp.n > 0; t++) {
      const int k = (int)((long)t * (sp.n - 1) / 8);
      if (sp.first[(size_t)k + 1] == sp.first[(size_t)k]) continue;
      const int nd = numa_node_of(sp.base[k]);
      if (nd >= 0 && nd < 8) { votes[nd]++; asked++; }
    }
    int nd = 0;
    for (int q = 1; q < 8; q++) if (votes[q] > votes[nd]) nd = q;
    if (asked > 0 && 3 * votes[nd] >= 2 * asked) {      // one node holds (most of) them: follow it
      std::lock_guard<std::mutex> lk(mu);
      bool found = false;
      for (auto &kv : known) if (kv.first == nd) { src_cpus = kv.second; found = true; }
      if (!found) { src_cpus = numa_node_cpus(nd); known.emplace_back(nd, src_cpus); }
    }
  }
  return staged_upload(ring, device, stream, d_dst, bytes, 48, [&sp, rec](char *dst, size_t off, size_t len) {
    sp.gather(dst, (long)(off / rec), (long)(len / rec));
  }, src_cpus.valid ? &src_cpus : nullptr);
}

}  // namespace balm
