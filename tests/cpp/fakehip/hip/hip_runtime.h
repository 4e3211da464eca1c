// A stand-in for <hip/hip_runtime.h> that lets tests/cpp/host_stage_test.cpp run balm_amd/csrc/host_stage.h on a box
// without a GPU: "device" memory is host memory, and a hipMemcpyAsync is DEFERRED -- it is executed (by a DMA thread) a little
// later, so a staging buffer that is refilled before its event was waited for corrupts the copy and the test sees it.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1 };
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
struct FakeEvent { std::atomic<long> ticket{0}; };
typedef FakeEvent *hipEvent_t;
typedef void *hipStream_t;

struct FakeDma {
  struct Op { void *dst; const void *src; size_t n; long ticket; };
  std::deque<Op> q;
  std::mutex mu;
  std::condition_variable cv;
  long issued = 0;
  std::atomic<long> completed{0};
  bool quit = false;
  std::thread th;
  FakeDma() : th([this] { run(); }) {}
  ~FakeDma() { { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); th.join(); }
  void run() {
    for (;;) {
      Op op;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit || !q.empty(); });
        if (q.empty()) return;
        op = q.front(); q.pop_front();
      }
      std::this_thread::sleep_for(std::chrono::microseconds(300));      // the copy happens LATE
      std::memcpy(op.dst, op.src, op.n);
      completed.store(op.ticket, std::memory_order_release);
    }
  }
  static FakeDma &get() { static FakeDma d; return d; }
};

inline hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorInvalidValue; }      // (no device: nobody is pinned)
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new FakeEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) {
  FakeDma &d = FakeDma::get();
  { std::lock_guard<std::mutex> lk(d.mu); d.q.push_back({dst, src, n, ++d.issued}); }
  d.cv.notify_all();
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  FakeDma &d = FakeDma::get();
  std::lock_guard<std::mutex> lk(d.mu);
  e->ticket.store(d.issued);
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) {
  FakeDma &d = FakeDma::get();
  while (d.completed.load(std::memory_order_acquire) < e->ticket.load()) std::this_thread::yield();
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) {
  FakeDma &d = FakeDma::get();
  long want; { std::lock_guard<std::mutex> lk(d.mu); want = d.issued; }
  while (d.completed.load(std::memory_order_acquire) < want) std::this_thread::yield();
  return hipSuccess;
}
