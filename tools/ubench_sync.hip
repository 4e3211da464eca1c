// Latency microbenchmarks behind the design of the persistent LDL^T solve (csrc/kernels_solve.hip) on gfx950:
//   1. flag ping-pong between two workgroups (same XCD / different XCD), with and without a 37 KB payload
//   2. dependent chains: f64 reciprocal (v_rcp_f64 + 2 Newton steps), f64 FMA, f64 MFMA 16x16x4
//   3. LDS publish -> barrier -> read-back round trip in a 256-thread workgroup
//   4. grid-wide barrier (atomic counter) over G co-resident workgroups
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_sync tools/ubench_sync.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int ld_acq(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_rlx(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// workgroup `a` and workgroup `b` bounce a flag `iters` times; payload doubles are written by the sender before the
// flag and summed by the receiver after it (0 = flag only).  out[0] = wall-clock ticks (100 MHz), out[1] = checksum
__global__ __launch_bounds__(256) void k_pingpong(int a, int b, int iters, int payload, int *flags, double *buf, long long *out) {
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  int *fa = flags, *fb = flags + 64;
  double *pa = buf, *pb = buf + payload + 64;
  double acc = 0;
  __shared__ double red[256];
  long long t0 = 0;
  if (me == a) {
    t0 = wall_clock64();
    for (int i = 1; i <= iters; i++) {
      for (int k = threadIdx.x; k < payload; k += 256) pa[k] = (double)(i + k);
      __syncthreads();
      if (threadIdx.x == 0) { st_rel(fa, i); while (ld_acq(fb) < i) __builtin_amdgcn_s_sleep(1); }
      __syncthreads();
      for (int k = threadIdx.x; k < payload; k += 256) acc += pb[k];
    }
  } else {
    for (int i = 1; i <= iters; i++) {
      if (threadIdx.x == 0) { while (ld_acq(fa) < i) __builtin_amdgcn_s_sleep(1); }
      __syncthreads();
      for (int k = threadIdx.x; k < payload; k += 256) acc += pa[k];
      for (int k = threadIdx.x; k < payload; k += 256) pb[k] = (double)(2 * i + k);
      __syncthreads();
      if (threadIdx.x == 0) st_rel(fb, i);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0; for (int k = 0; k < 256; k++) s += red[k];
    if (me == a) { out[0] = wall_clock64() - t0; out[1] = (long long)s; } else out[2] = (long long)s;
  }
}

__device__ __forceinline__ double rcp_nr(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return x;
}

// mode 0: rcp_nr chain; 1: fma chain; 2: dependent MFMA chain; 3: 4 independent MFMA accumulators; 4: rcp only; 5: rcp + 1 Newton step
__global__ __launch_bounds__(64) void k_chain(int mode, int iters, double seed, double *out, long long *ticks) {
  double x = seed + threadIdx.x * 1e-9;
  d4 acc = {0, 0, 0, 0}, acc1 = acc, acc2 = acc, acc3 = acc;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  if (mode == 0) for (int i = 0; i < iters; i++) x = rcp_nr(x) + 0.5;
  else if (mode == 1) for (int i = 0; i < iters; i++) x = __builtin_fma(x, 0.999999, 1e-7);
  else if (mode == 2) for (int i = 0; i < iters; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
  else if (mode == 3) for (int i = 0; i < iters; i++) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc3, 0, 0, 0);
  } else if (mode == 4) for (int i = 0; i < iters; i++) x = __builtin_amdgcn_rcp(x) + 0.5;
  else if (mode == 5) for (int i = 0; i < iters; i++) { double y = __builtin_amdgcn_rcp(x); y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y); x = y + 0.5; }
  else if (mode == 6) for (int i = 0; i < iters; i++) {       // MFMA -> VALU read of the result -> next MFMA operand
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    x = acc[0] * 1e-30 + seed;
  }
  const long long c1 = clock64();
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x + acc[0] + acc1[1] + acc2[2] + acc3[3];
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}

// LDS round trip: every thread writes a value, barrier, thread reads 4 broadcast values + its own row, barrier
__global__ __launch_bounds__(256) void k_lds(int iters, double *out, long long *ticks) {
  __shared__ double cur[4][128];
  double x = threadIdx.x * 1e-3;
  const long long t0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    cur[threadIdx.x >> 6][threadIdx.x & 63] = x;
    __syncthreads();
    x = cur[0][i & 63] + cur[1][(i + 1) & 63] + cur[2][(i + 2) & 63] + cur[3][threadIdx.x & 63];
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

// grid barrier: monotone counter, every workgroup adds 1 and spins until the generation's total is reached
__global__ __launch_bounds__(256) void k_gridbar(int iters, int *ctr, long long *ticks) {
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters; i++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const int want = i * (int)gridDim.x;
      while (ld_acq(ctr) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = wall_clock64() - t0;
}

int main() {
  int *flags; double *buf, *out; long long *ticks;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&buf, 1 << 22)); CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&ticks, 64));
  long long h[8];
  const double TICK_US = 0.01;   // wall_clock64: 100 MHz
  printf("# flag ping-pong, round trip = 2 hops; workgroup b = 8 shares workgroup 0's XCD (round-robin dispatch), b = 1..7 do not\n");
  for (int payload : {0, 4608, 18432}) {
    for (int b : {8, 1, 2, 4, 16, 9}) {
      CK(hipMemset(flags, 0, 4096));
      const int iters = 2000;
      hipLaunchKernelGGL(k_pingpong, dim3(32), dim3(256), 0, 0, 0, b, iters, payload, flags, buf, ticks);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, ticks, 24, hipMemcpyDeviceToHost));
      printf("pingpong wg0<->wg%-2d payload %6d B: %.3f us per round trip (%.3f us per hop)\n", b, payload * 8, h[0] * TICK_US / iters,
             h[0] * TICK_US / iters / 2);
    }
  }
  const char *names[] = {"rcp_f64 + 2 Newton (+1 add)", "fma_f64", "mfma_f64_16x16x4 dependent", "mfma_f64 x4 independent (per 4)",
                         "rcp_f64 only (+1 add)", "rcp_f64 + 1 Newton (+1 add)", "mfma -> valu -> mfma"};
  for (int mode = 0; mode < 7; mode++) {
    const int iters = 20000;
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, mode, iters, 1.37, out, ticks);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost));
    printf("chain %-34s: %.1f ns per link (%.1f clock64 ticks)\n", names[mode], h[0] * TICK_US * 1e3 / iters, (double)h[1] / iters);
  }
  {
    const int iters = 20000;
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 0, 0, iters, out, ticks);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost));
    printf("lds publish/barrier/read/barrier (256 threads): %.1f ns per round\n", h[0] * TICK_US * 1e3 / iters);
  }
  for (int G : {8, 32, 64, 128, 256}) {
    CK(hipMemset(flags, 0, 4096));
    const int iters = 1000;
    hipLaunchKernelGGL(k_gridbar, dim3(G), dim3(256), 0, 0, iters, flags, ticks);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, ticks, 8, hipMemcpyDeviceToHost));
    printf("grid barrier over %3d workgroups: %.3f us\n", G, h[0] * TICK_US / iters);
  }
  return 0;
}
