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
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace balm {

class HostPool {
 public:
  static HostPool &get() { static HostPool p; return p; }
  int workers() const { return (int)th_.size(); }
  // every pool thread (and the caller) runs fn(thread index) once; returns when all have returned.  One job at a time.
  void run_all(const std::function<void(int)> &fn) {
    std::lock_guard<std::mutex> job(job_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; pending_ = (int)th_.size(); gen_++;
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
    for (;;) {
      const std::function<void(int)> *f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_go_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_; f = fn_;
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
  unsigned long gen_ = 0;
  int pending_ = 0;
  bool quit_ = false;
};

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

  hipError_t init() {
    if (buf[0]) return hipSuccess;
    for (int b = 0; b < NBUF; b++) {
      hipError_t e = hipHostMalloc((void **)&buf[b], CHUNK, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
      if (e != hipSuccess) { release(); return e; }
    }
    return hipSuccess;
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
inline hipError_t staged_upload(PinnedRing &ring, hipStream_t stream, void *d_dst, size_t bytes, size_t unit,
                                const std::function<void(char *, size_t, size_t)> &fill) {
  if (bytes == 0) return hipSuccess;
  hipError_t e = ring.init();
  if (e != hipSuccess) return e;
  if (unit == 0 || unit > PinnedRing::CHUNK) return hipErrorInvalidValue;
  // the buffers may still feed the DMAs of an earlier upload
  for (int b = 0; b < PinnedRing::NBUF; b++)
    if (ring.busy[b]) { (void)hipEventSynchronize(ring.ev[b]); ring.busy[b] = false; }
  if (bytes <= ((size_t)1 << 20)) {      // small: the calling thread alone, no pool wake-up
    fill(ring.buf[0], 0, bytes);
    e = hipMemcpyAsync(d_dst, ring.buf[0], bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(ring.ev[0], stream);
    if (e == hipSuccess) ring.busy[0] = true;
    return e;
  }
  HostPool &pool = HostPool::get();
  const size_t chunk = PinnedRing::CHUNK / unit * unit;
  const long nchunks = (long)((bytes + chunk - 1) / chunk);
  // slices: about 256 KiB each so that the threads finish a chunk together, a multiple of the unit
  size_t slice = ((size_t)256 << 10) / unit * unit;
  if (slice == 0) slice = unit;
  const long per_chunk = (long)((chunk + slice - 1) / slice);
  std::vector<std::atomic<int>> done((size_t)nchunks);
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  std::atomic<long> next{0}, allowed{PinnedRing::NBUF - 1};
  std::atomic<bool> abort{false};
  auto slices_of = [&](long k) {
    const size_t len = std::min(chunk, bytes - (size_t)k * chunk);
    return (long)((len + slice - 1) / slice);
  };
  hipError_t err = hipSuccess;
  auto worker = [&](int) {
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
      fill(ring.buf[k % PinnedRing::NBUF] + (size_t)s * slice, off, len);
      done[(size_t)k].fetch_add(1, std::memory_order_release);
    }
  };
  auto driver = [&](int tid) {
    if (tid != pool.workers()) { worker(tid); return; }
    // the calling thread: DMA of chunk k as soon as its slices are in, buffer of chunk k - 1 freed behind it; fills
    // slices itself while it has nothing to issue
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
              fill(ring.buf[kk % PinnedRing::NBUF] + (size_t)s * slice, off, len);
              done[(size_t)kk].fetch_add(1, std::memory_order_release);
            }
          }
        } else {
          std::this_thread::yield();
        }
      }
      const int b = (int)(k % PinnedRing::NBUF);
      const size_t len = std::min(chunk, bytes - (size_t)k * chunk);
      hipError_t e2 = hipMemcpyAsync((char *)d_dst + (size_t)k * chunk, ring.buf[b], len, hipMemcpyHostToDevice, stream);
      if (e2 == hipSuccess) e2 = hipEventRecord(ring.ev[b], stream);
      if (e2 != hipSuccess) { err = e2; abort.store(true); allowed.store(nchunks, std::memory_order_release); return; }
      ring.busy[b] = true;
      if (k >= 1) {
        const int pb = (int)((k - 1) % PinnedRing::NBUF);
        e2 = hipEventSynchronize(ring.ev[pb]);
        if (e2 != hipSuccess) { err = e2; abort.store(true); allowed.store(nchunks, std::memory_order_release); return; }
        ring.busy[pb] = false;
        allowed.store(k - 1 + PinnedRing::NBUF, std::memory_order_release);
      }
    }
  };
  pool.run_all(driver);
  return err;
}

// the plain case: a contiguous caller array
inline hipError_t staged_copy(PinnedRing &ring, hipStream_t stream, void *d_dst, const void *src, size_t bytes, size_t unit = 64) {
  if (bytes < ((size_t)1 << 20))      // small: the runtime's own path (one bounce, no pool wake-up)
    return hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream);
  const char *s = static_cast<const char *>(src);
  return staged_upload(ring, stream, d_dst, bytes, unit, [s](char *dst, size_t off, size_t len) { std::memcpy(dst, s + off, len); });
}

}  // namespace balm
