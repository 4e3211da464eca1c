// How long is a chain of dependent FP64 additions on one wavefront?  (k_seg_wave's add phase: 64 per chunk and term column.)
// hipcc --offload-arch=gfx950 -O3 tools/ubench_addchain.hip -o tools/bin/ubench_addchain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(double *out, const double *in, int n, int mode) {
  double acc = in[threadIdx.x & 63];
  const double a = in[64], b = in[65];
  if (mode == 0) {                       // dependent adds, operands in registers
    for (int i = 0; i < n; i++) {
#pragma unroll
      for (int u = 0; u < 64; u++) acc = __dadd_rn(acc, (u & 1) ? a : b);
    }
  } else {                               // the same through LDS reads (64 values per round, prefetched 16 ahead like k_seg_wave)
    __shared__ double sm[4][64 * 65];
    double *mine = sm[threadIdx.x >> 6];
    for (int c = 0; c < 18; c++) mine[c * 65 + (threadIdx.x & 63)] = a + c;
    __syncthreads();
    const double *colp = mine + min((int)(threadIdx.x & 63), 17) * 65;
    for (int i = 0; i < n; i++) {
      if ((threadIdx.x & 63) < 18) {
        double va[16], vb[16];
#pragma unroll
        for (int u = 0; u < 16; u++) va[u] = colp[u];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (q < 3) {
#pragma unroll
            for (int u = 0; u < 16; u++) vb[u] = colp[16 * (q + 1) + u];
          }
#pragma unroll
          for (int u = 0; u < 16; u++) acc = __dadd_rn(acc, va[u]);
#pragma unroll
          for (int u = 0; u < 16; u++) va[u] = vb[u];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  double *in, *out;
  hipMalloc(&in, 1024); hipMalloc(&out, 8 * 1024 * 1024);
  double h[66]; for (int i = 0; i < 66; i++) h[i] = 1.0 + i * 1e-3;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 2000;
  for (int mode = 0; mode < 2; mode++)
    for (int blocks : {1, 256, 1024, 4096})
      for (int threads : {64, 256}) {
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, out, in, 10, mode);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, out, in, n, mode);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s, %4d workgroups x %d wavefront(s): %.1f ns per 64 dependent additions (%.2f ns each)\n", mode ? "LDS column reads + adds" : "adds on registers", blocks,
               threads / 64, ms * 1e6 / n, ms * 1e6 / n / 64);
      }
  return 0;
}
