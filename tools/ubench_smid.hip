// which ids does __smid() hand out on this device, and how expensive is a contended atomic counter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k_ids(unsigned *out) { if (threadIdx.x == 0) out[blockIdx.x] = __smid(); }
__global__ void k_queue(int *q, int ntiles, int chunk, long long *sink) {
  const int lane = threadIdx.x & 63;
  long long acc = 0;
  for (;;) {
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(q, chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= ntiles) break;
    acc += t;
  }
  if (acc == -1) *sink = acc;
}
int main() {
  unsigned *d; const int nb = 4096;
  hipMalloc(&d, nb * sizeof(unsigned));
  hipLaunchKernelGGL(k_ids, dim3(nb), dim3(256), 0, 0, d);
  std::vector<unsigned> h(nb);
  hipMemcpy(h.data(), d, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::set<unsigned> s(h.begin(), h.end());
  unsigned mx = 0; for (unsigned v : h) mx = v > mx ? v : mx;
  printf("__smid over %d blocks: %zu distinct ids, max %u; first 16:", nb, s.size(), mx);
  for (int i = 0; i < 16; i++) printf(" %u", h[i]);
  printf("\n");
  int *q; long long *sink; hipMalloc(&q, 4); hipMalloc(&sink, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int chunk : {1, 4, 16}) {
    hipMemset(q, 0, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_queue, dim3(768), dim3(256), 0, 0, q, 10000, chunk, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("10000 tiles from one counter, %d per fetch, 768 blocks x 4 waves: %.1f us\n", chunk, ms * 1e3);
  }
  return 0;
}
