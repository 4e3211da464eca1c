// Host -> HBM yardsticks for balm_amd/csrc/host_stage.h: what the pieces of the pinned-ring upload can do alone on this box.
//   1. hipMemcpyAsync pinned -> device, by chunk size; two streams at once
//   2. a kernel reading the pinned chunk over the link itself (mapped host memory)
//   3. host memcpy pageable -> pinned by thread count
//   4. hipMemcpyAsync straight from pageable memory (what round 4's entry points did)
// hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench_h2d.hip -o tools/bin/ubench_h2d
//   5. the library's own pipeline (balm_amd/csrc/host_stage.h) on the same buffer: -DBALM_HOST_POOL_THREADS / _STAGE_CHUNK_MB / _STAGE_NBUF
#include <hip/hip_runtime.h>
#include "../balm_amd/csrc/host_stage.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_pull(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t total = (size_t)512 << 20;
  char *pin = nullptr, *dev = nullptr;
  CK(hipHostMalloc((void **)&pin, total, hipHostMallocDefault));
  CK(hipMalloc((void **)&dev, total));
  std::vector<char> page(total);
  memset(pin, 1, total); memset(page.data(), 2, total);
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipMemcpy(dev, pin, total, hipMemcpyHostToDevice));
  for (size_t chunk : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, total}) {
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      const double t0 = now();
      for (size_t off = 0; off < total; off += chunk) CK(hipMemcpyAsync(dev + off, pin + off, chunk, hipMemcpyHostToDevice, s0));
      CK(hipStreamSynchronize(s0));
      const double gbs = total / (now() - t0) / 1e9;
      if (gbs > best) best = gbs;
    }
    printf("pinned -> device, one stream, %4zu MB chunks: %6.1f GB/s\n", chunk >> 20, best);
  }
  {
    double best = 0;
    const size_t chunk = (size_t)16 << 20;
    for (int rep = 0; rep < 3; rep++) {
      const double t0 = now();
      int k = 0;
      for (size_t off = 0; off < total; off += chunk, k++) CK(hipMemcpyAsync(dev + off, pin + off, chunk, hipMemcpyHostToDevice, (k & 1) ? s1 : s0));
      CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
      const double gbs = total / (now() - t0) / 1e9;
      if (gbs > best) best = gbs;
    }
    printf("pinned -> device, two streams alternating, 16 MB chunks: %6.1f GB/s\n", best);
  }
  {
    char *pin_dev = nullptr;
    CK(hipHostGetDevicePointer((void **)&pin_dev, pin, 0));
    for (int grid : {256, 1024, 4096}) {
      double best = 0;
      for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        hipLaunchKernelGGL(k_pull, dim3(grid), dim3(256), 0, s0, (const float4 *)pin_dev, (float4 *)dev, total / 16);
        CK(hipStreamSynchronize(s0));
        const double gbs = total / (now() - t0) / 1e9;
        if (gbs > best) best = gbs;
      }
      printf("kernel pulls the pinned buffer over the link, %4d workgroups: %6.1f GB/s\n", grid, best);
    }
  }
  for (int T : {1, 2, 4, 8, 16, 32}) {
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      const double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] { const size_t a = total * t / T, b = total * (t + 1) / T; memcpy(pin + a, page.data() + a, b - a); });
      for (auto &x : th) x.join();
      const double gbs = total / (now() - t0) / 1e9;
      if (gbs > best) best = gbs;
    }
    printf("host memcpy pageable -> pinned, %2d threads: %6.1f GB/s\n", T, best);
  }
  {
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
      const double t0 = now();
      CK(hipMemcpyAsync(dev, page.data(), total, hipMemcpyHostToDevice, s0));
      CK(hipStreamSynchronize(s0));
      const double gbs = total / (now() - t0) / 1e9;
      if (gbs > best) best = gbs;
    }
    printf("hipMemcpyAsync straight from pageable memory: %6.1f GB/s\n", best);
  }
  {   // a FRESH pageable buffer every time (what a caller's first upload of a new array sees)
    for (int rep = 0; rep < 3; rep++) {
      std::vector<char> fresh(total, (char)rep);
      const double t0 = now();
      CK(hipMemcpyAsync(dev, fresh.data(), total, hipMemcpyHostToDevice, s0));
      CK(hipStreamSynchronize(s0));
      printf("hipMemcpyAsync from a fresh pageable buffer, rep %d: %6.1f GB/s\n", rep, total / (now() - t0) / 1e9);
    }
  }
  {
    balm::PinnedRing ring;
    for (int rep = 0; rep < 4; rep++) {
      std::vector<char> fresh(total, (char)rep);
      const double t0 = now();
      CK(balm::staged_copy(ring, 0, s0, dev, fresh.data(), total));
      const double t1 = now();
      CK(hipStreamSynchronize(s0));
      printf("host_stage.h pipeline (%d threads, %d x %d MB), fresh buffer, rep %d: %6.1f GB/s  (host side done after %.2f of %.2f ms)\n",
             balm::HostPool::get().workers() + 1, BALM_STAGE_NBUF, BALM_STAGE_CHUNK_MB, rep, total / (now() - t0) / 1e9, (t1 - t0) * 1e3, (now() - t0) * 1e3);
    }
    for (int rep = 0; rep < 2; rep++) {
      const double t0 = now();
      CK(balm::staged_copy(ring, 0, s0, dev, page.data(), total));
      CK(hipStreamSynchronize(s0));
      printf("host_stage.h pipeline, the warm buffer, rep %d: %6.1f GB/s\n", rep, total / (now() - t0) / 1e9);
    }
    ring.release();
  }
  printf("host threads: %u\n", std::thread::hardware_concurrency());
  return 0;
}
