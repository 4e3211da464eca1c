// The strided point entries' host side on this box (balm_amd/csrc/host_stage.h StridedPoints / staged_points): 177 clouds of 48-byte
// elements shaped like the shipped window, packed to 12-byte xyz through the pinned ring.  By pool size (-DBALM_HOST_POOL_THREADS,
// -DBALM_STAGE_FILL_THREADS), software prefetch distance (-DBALM_GATHER_PREFETCH) and by WHERE the caller's clouds live: first touched by
// an unpinned main thread, by a thread on the GPU's own NUMA node, by a thread on another node, half and half.  Also: what PinnedRing::init costs.
// hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench_gather.hip -o tools/bin/ubench_gather
#include <hip/hip_runtime.h>
#include "../balm_amd/csrc/host_stage.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int W = 177; const long per = 75736;
  CK(hipSetDevice(0));
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  float *dev = nullptr;
  CK(hipMalloc((void **)&dev, (size_t)W * per * 12 + 64));
  {   // the ring's first use, piece by piece
    double t0 = now();
    char *b[3];
    for (int k = 0; k < 3; k++) CK(hipHostMalloc((void **)&b[k], (size_t)32 << 20, hipHostMallocDefault));
    double t1 = now();
    for (int k = 0; k < 3; k++) memset(b[k], 0, (size_t)32 << 20);
    double t2 = now();
    printf("3 x 32 MB hipHostMalloc %.2f ms, memset %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    for (int k = 0; k < 3; k++) CK(hipHostFree(b[k]));
    t0 = now();
    char *one; CK(hipHostMalloc((void **)&one, (size_t)96 << 20, hipHostMallocDefault));
    printf("1 x 96 MB hipHostMalloc %.2f ms\n", (now() - t0) * 1e3);
    CK(hipHostFree(one));
  }
  balm::GpuNode node = balm::gpu_local_cpus(0);
  cpu_set_t all, far;
  sched_getaffinity(0, sizeof(all), &all);
  CPU_ZERO(&far);
  int nfar = 0;
  for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &all) && !(node.valid && CPU_ISSET(c, &node.cpus))) { CPU_SET(c, &far); nfar++; }
  printf("gpu-local cpus known: %d, other cpus: %d, pool %d threads\n", (int)node.valid, nfar, balm::HostPool::get().workers() + 1);
  balm::PinnedRing ring;
  for (int where = 0; where < 4; where++) {      // 3: the first half of the clouds touched on the GPU's node, the second on another (a reader thread moved between sockets)
    if (where == 1 && !node.valid) continue;
    if (where >= 2 && nfar == 0) continue;
    if (where == 3 && !node.valid) continue;
    if (where == 1) sched_setaffinity(0, sizeof(cpu_set_t), &node.cpus);
    if (where == 2) sched_setaffinity(0, sizeof(cpu_set_t), &far);
    if (where == 0) sched_setaffinity(0, sizeof(cpu_set_t), &all);
    std::vector<std::vector<char>> clouds((size_t)W);
    std::vector<const void *> base((size_t)W);
    std::vector<long> cnt((size_t)W, per);
    for (int k = 0; k < W; k++) {
      if (where == 3) sched_setaffinity(0, sizeof(cpu_set_t), k < W / 2 ? &node.cpus : &far);
      clouds[(size_t)k].assign((size_t)per * 48, (char)(k + 1)); base[(size_t)k] = clouds[(size_t)k].data();
    }
    sched_setaffinity(0, sizeof(cpu_set_t), &all);
    balm::StridedPoints sp;
    sp.set(W, base.data(), cnt.data(), 48);
    {
      int hist[16] = {0};
      for (int k = 0; k < W; k++) { const int nd = balm::numa_node_of(base[(size_t)k]); if (nd >= 0 && nd < 16) hist[nd]++; else hist[15]++; }
      printf("  clouds by NUMA node of their first page:");
      for (int q = 0; q < 16; q++) if (hist[q]) printf(" node%d=%d", q, hist[q]);
      printf("  (main thread now on cpu %d)\n", sched_getcpu());
    }
    const double bytes = (double)sp.total() * 12;
    for (int rep = 0; rep < 6; rep++) {
      const double t0 = now();
      CK(balm::staged_points(ring, 0, s0, dev, sp));
      const double t1 = now();
      CK(hipStreamSynchronize(s0));
      const double t2 = now();
      printf("clouds first touched %s: rep %d  host side %.2f ms, arrived %.2f ms = %.1f GB/s packed (%.0f GB/s read off the clouds)\n",
             where == 0 ? "by the unpinned main thread" : where == 1 ? "on the GPU's node" : where == 2 ? "on ANOTHER node" : "HALF on the GPU's node, half on another", rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3,
             bytes / (t2 - t0) / 1e9, 4 * bytes / (t2 - t0) / 1e9);
    }
  }
  ring.release();
  return 0;
}
