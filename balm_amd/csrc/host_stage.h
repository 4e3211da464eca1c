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
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

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

class HostPool {
 public:
  static HostPool &get() { static HostPool p; return p; }
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
  HostPool() {
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc >= 64 ? 15 : hc >= 16 ? 7 : hc >= 4 ? 3 : 1;      // + the calling thread
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
        if (pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &aff->cpus) == 0) { cur_aff = aff->cpus; have_aff = true; }
      }
      (*f)(t);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) cv_done_.notify_all();
      }
    }
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

// fn(lo, hi) over [0, n) in contiguous pieces, one or a few per pool thread; serial below `grain`
inline void parallel_ranges(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn) {
  if (n == 0) return;
  HostPool &pool = HostPool::get();
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

  GpuNode node;                // the CPUs next to the device the ring feeds (invalid: unknown, nobody is pinned)
  int pos = 0;                 // buffer of the next upload's first chunk: an upload continues round the ring where the previous one stopped,
                               // so its first chunks fill buffers that are already free while the previous upload's last DMAs still run

  // allocated (and first touched) by a pool thread that sits on the device's own node
  hipError_t init(int device) {
    if (buf[0]) return hipSuccess;
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
    HostPool &pool = HostPool::get();
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
                                const std::function<void(char *, size_t, size_t)> &fill) {
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
  HostPool &pool = HostPool::get();
  const long p0 = ring.pos;
  const size_t chunk = PinnedRing::CHUNK / unit * unit;
  const long nchunks = (long)((bytes + chunk - 1) / chunk);
  // slices: about 256 KiB each so that the threads finish a chunk together, a multiple of the unit
  size_t slice = ((size_t)256 << 10) / unit * unit;
  if (slice == 0) slice = unit;
  const long per_chunk = (long)((chunk + slice - 1) / slice);
  std::vector<std::atomic<int>> done((size_t)nchunks);
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  std::atomic<long> next{0}, allowed{-1};      // chunks [0, allowed] may be filled: their buffers are free
  std::atomic<bool> abort{false};
  auto slices_of = [&](long k) {
    const size_t len = std::min(chunk, bytes - (size_t)k * chunk);
    return (long)((len + slice - 1) / slice);
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
      const long k = t / per_chunk, s = t % per_chunk;
      if (k >= nchunks) return;
      if (s >= slices_of(k)) continue;
      while (allowed.load(std::memory_order_acquire) < k) {
        if (abort.load(std::memory_order_relaxed)) return;
        std::this_thread::yield();
      }
      const size_t off = (size_t)k * chunk + (size_t)s * slice;
      const size_t len = std::min(slice, std::min(chunk, bytes - (size_t)k * chunk) - (size_t)s * slice);
      fill(ring.buf[(p0 + k) % PinnedRing::NBUF] + (size_t)s * slice, off, len);
      done[(size_t)k].fetch_add(1, std::memory_order_release);
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
        if (t / per_chunk <= allowed.load(std::memory_order_relaxed) && t / per_chunk < nchunks) {
          long mine = t;
          if (next.compare_exchange_strong(mine, t + 1, std::memory_order_relaxed)) {
            const long kk = t / per_chunk, s = t % per_chunk;
            if (s < slices_of(kk)) {
              const size_t off = (size_t)kk * chunk + (size_t)s * slice;
              const size_t len = std::min(slice, std::min(chunk, bytes - (size_t)kk * chunk) - (size_t)s * slice);
              fill(ring.buf[(p0 + kk) % PinnedRing::NBUF] + (size_t)s * slice, off, len);
              done[(size_t)kk].fetch_add(1, std::memory_order_release);
            }
          }
        } else {
          std::this_thread::yield();
        }
      }
      const int b = (int)((p0 + k) % PinnedRing::NBUF);
      const size_t len = std::min(chunk, bytes - (size_t)k * chunk);
#ifdef BALM_STAGE_TRACE
      tr_ready[(size_t)k] = tr_now();
#endif
      hipError_t e2 = stage_enqueue((char *)d_dst + (size_t)k * chunk, ring.buf[b], len, stream);
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
  pool.run_all(driver, &ring.node);
  ring.pos = (int)((p0 + nchunks) % PinnedRing::NBUF);
#ifdef BALM_STAGE_TRACE
  fprintf(stderr, "[stage] %zu bytes, %ld chunks of %zu MB, %d threads; per chunk: filled / DMA issued / previous chunk's DMA seen done (ms)\n", bytes, nchunks,
          chunk >> 20, pool.workers() + 1);
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

}  // namespace balm
