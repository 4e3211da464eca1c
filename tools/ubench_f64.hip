// FP64 peak microbenchmarks for gfx950: v_mfma_f64_16x16x4_f64, v_fma_f64, and both together.
// Confirms the roofline peak that bench.py prices hessian_syrk against (MI355X_MICROARCH.md gives
// 157.3 TF fp32 vector/matrix; FP64 = half that = 78.6 TF is the working assumption).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// f64 MFMA rate with the accumulators pinned to physical AGPRs (what k_hessian_syrk does): the builtin form keeps
// loop-carried accumulators in arch VGPRs and copies them to and from AGPRs around every MFMA, which
// measures the copies, not the matrix pipe.  NACC = 8 accumulator tiles a[0:7] .. a[56:63].
#define UB_CLOB "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double seed) {
  static_assert(NACC == 8, "eight pinned accumulator tiles");
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
#pragma unroll
  for (int r = 0; r < 64; r += 8)
    asm volatile("v_accvgpr_write_b32 a0, 0" ::: UB_CLOB);        // (zero start values do not matter for the rate)
  for (int it = 0; it < iters; it++) {
    asm volatile(
        "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n\t"
        "v_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n\t"
        "v_mfma_f64_16x16x4_f64 a[16:23], %0, %1, a[16:23]\n\t"
        "v_mfma_f64_16x16x4_f64 a[24:31], %0, %1, a[24:31]\n\t"
        "v_mfma_f64_16x16x4_f64 a[32:39], %0, %1, a[32:39]\n\t"
        "v_mfma_f64_16x16x4_f64 a[40:47], %0, %1, a[40:47]\n\t"
        "v_mfma_f64_16x16x4_f64 a[48:55], %0, %1, a[48:55]\n\t"
        "v_mfma_f64_16x16x4_f64 a[56:63], %0, %1, a[56:63]"
        :: "v"(a), "v"(b) : UB_CLOB);
  }
  unsigned lo, hi;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1" : "=v"(lo), "=v"(hi) :: UB_CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = __hiloint2double(hi, lo);
}

template <int NCH>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double seed) {
  double x[NCH];
  for (int i = 0; i < NCH; i++) x[i] = seed + i + threadIdx.x * 1e-6;
  const double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NCH; i++) x[i] = __builtin_fma(x[i], m, c);
  }
  double s = 0;
  for (int i = 0; i < NCH; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0..(nm-1) of each block run MFMA, the rest VALU fma: do the two pipes overlap for f64?
template <int NACC, int NCH>
__global__ __launch_bounds__(512) void k_mixed(double *out, int iters, int iters_fma, int n_mfma_waves, double seed) {
  const int wv = threadIdx.x >> 6;
  double s = 0;
  if (wv < n_mfma_waves) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[NCH];
    for (int i = 0; i < NCH; i++) x[i] = seed + i + threadIdx.x * 1e-6;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters_fma; it++) {
#pragma unroll
      for (int i = 0; i < NCH; i++) x[i] = __builtin_fma(x[i], m, c);
    }
    for (int i = 0; i < NCH; i++) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_copy(const double4 *__restrict__ in, double4 *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// HBM yardsticks (round 4: VERDICT r3 item 6 -- the double4 grid-stride copy above reaches 4.94 TB/s, the guide's float4 copy 6.29):
// 16-byte accesses, U independent accesses per lane in flight, streaming (nontemporal) variants, and the read-only / write-only rates
// (K2 is 36 % reads, 64 % writes).
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy16(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * stride < n) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * stride < n) { if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
  }
}
template <int U>
__global__ __launch_bounds__(256) void k_read16(const f4 *__restrict__ in, float *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; u++) if (i + u * stride < n) acc += in[i + u * stride];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_fill16(f4 *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}

template <class F>
float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f();  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  const int CU = p.multiProcessorCount;
  double *out; hipMalloc(&out, sizeof(double) * 512 * CU * 8 * 4);
  const int iters = 20000;
  // blocks of 4 wavefronts (the kernels' launch bound); k blocks per CU = 4 k wavefronts per CU = k per SIMD.
  // One wavefront per SIMD cannot keep the f64 MFMA pipe full (dependent-issue latency), two or more can.
  auto launched = [](const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", what, hipGetErrorString(e)); exit(1); }
  };
  for (int per_simd : {1, 2, 4}) {
    const int wpc = 4 * per_simd;
    {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(CU * per_simd), dim3(256), 0, 0, out, iters, 1.0); launched("k_mfma"); });
      double fl = (double)CU * wpc * iters * 8 * 2048.0;
      printf("mfma_f64_16x16x4  %2d waves/CU: %8.3f ms  %7.2f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wpc, ms, fl / ms / 1e9,
             ms * 1e-3 * 2.4e9 / ((double)iters * 8 * per_simd));
    }
    {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_fma<16>, dim3(CU * per_simd), dim3(256), 0, 0, out, iters, 1.0); launched("k_fma"); });
      double fl = (double)CU * wpc * 64 * iters * 16 * 2.0;
      printf("v_fma_f64         %2d waves/CU: %8.3f ms  %7.2f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
    }
  }
  {  // 8 waves/CU: 4 MFMA + 4 VALU, sized to take about equally long alone
    const int im = iters, ifma = iters * 4;     // 8 MFMA*2048 flop vs 16 fma*128 flop per iteration
    float ms = time_ms([&] { hipLaunchKernelGGL((k_mixed<8, 16>), dim3(CU), dim3(512), 0, 0, out, im, ifma, 4, 1.0); });
    double fm = (double)CU * 4 * im * 8 * 2048.0, fv = (double)CU * 4 * 64 * (double)ifma * 16 * 2.0;
    printf("mixed 4 MFMA + 4 VALU waves/CU: %8.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", ms, fm / ms / 1e9, fv / ms / 1e9,
           (fm + fv) / ms / 1e9);
    float ms_m = time_ms([&] { hipLaunchKernelGGL((k_mixed<8, 16>), dim3(CU), dim3(512), 0, 0, out, im, 0, 4, 1.0); });
    float ms_v = time_ms([&] { hipLaunchKernelGGL((k_mixed<8, 16>), dim3(CU), dim3(512), 0, 0, out, 0, ifma, 4, 1.0); });
    printf("   alone: mfma part %.3f ms, valu part %.3f ms  (overlap if mixed ~ max, serial if ~ sum)\n", ms_m, ms_v);
  }
  {
    size_t n = (size_t)1 << 26;   // 64M double4 = 2 GiB
    double4 *a, *b; hipMalloc(&a, n * 32); hipMalloc(&b, n * 32); hipMemset(a, 1, n * 32);
    float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(CU * 8), dim3(256), 0, 0, a, b, n); });
    printf("copy 2 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 32 / ms / 1e9);
    // 16-byte variants over the same 2 GiB: grid = blocks per CU x CUs, U accesses per lane in flight
    const size_t n16 = n * 2;
    const f4 *a16 = (const f4 *)a; f4 *b16 = (f4 *)b;
    double best = 0.0;
    auto report = [&](const char *what, float t, double bytes) {
      const double tbs = bytes / t / 1e9;
      printf("  %-58s %.3f ms  %.2f TB/s\n", what, t, tbs);
      return tbs;
    };
    for (int bpc : {8, 16, 32}) {
      char nm[96];
      float t;
      t = time_ms([&] { hipLaunchKernelGGL((k_copy16<1, false>), dim3(CU * bpc), dim3(256), 0, 0, a16, b16, n16); });
      snprintf(nm, sizeof nm, "copy float4, %d blocks/CU, 1 in flight", bpc); best = fmax(best, report(nm, t, 2.0 * n * 32));
      t = time_ms([&] { hipLaunchKernelGGL((k_copy16<4, false>), dim3(CU * bpc), dim3(256), 0, 0, a16, b16, n16); });
      snprintf(nm, sizeof nm, "copy float4, %d blocks/CU, 4 in flight", bpc); best = fmax(best, report(nm, t, 2.0 * n * 32));
      t = time_ms([&] { hipLaunchKernelGGL((k_copy16<4, true>), dim3(CU * bpc), dim3(256), 0, 0, a16, b16, n16); });
      snprintf(nm, sizeof nm, "copy float4, %d blocks/CU, 4 in flight, nontemporal", bpc); best = fmax(best, report(nm, t, 2.0 * n * 32));
      t = time_ms([&] { hipLaunchKernelGGL((k_copy16<8, true>), dim3(CU * bpc), dim3(256), 0, 0, a16, b16, n16); });
      snprintf(nm, sizeof nm, "copy float4, %d blocks/CU, 8 in flight, nontemporal", bpc); best = fmax(best, report(nm, t, 2.0 * n * 32));
    }
    {
      float t = time_ms([&] { hipLaunchKernelGGL((k_copy16<1, false>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, a16, b16, n16); });
      best = fmax(best, report("copy float4, one element per thread (8M blocks)", t, 2.0 * n * 32));
    }
    printf("best copy rate: %.2f TB/s (read+write)\n", best);
    {
      float t = time_ms([&] { hipLaunchKernelGGL((k_read16<8>), dim3(CU * 16), dim3(256), 0, 0, a16, (float *)out, n16); });
      report("read-only float4, 16 blocks/CU, 8 in flight", t, 1.0 * n * 32);
      t = time_ms([&] { hipLaunchKernelGGL((k_fill16<false>), dim3(CU * 16), dim3(256), 0, 0, b16, n16); });
      report("write-only float4, 16 blocks/CU", t, 1.0 * n * 32);
      t = time_ms([&] { hipLaunchKernelGGL((k_fill16<true>), dim3(CU * 16), dim3(256), 0, 0, b16, n16); });
      report("write-only float4, 16 blocks/CU, nontemporal", t, 1.0 * n * 32);
      // (the copy is fastest with one element per thread -- dispatch order = address order; is the write rate a property of the grid-stride form?)
      t = time_ms([&] { hipLaunchKernelGGL((k_fill16<false>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, b16, n16); });
      report("write-only float4, one element per thread (8M blocks)", t, 1.0 * n * 32);
      t = time_ms([&] { hipLaunchKernelGGL((k_fill16<true>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, b16, n16); });
      report("write-only float4, one element per thread, nontemporal", t, 1.0 * n * 32);
      for (int bpc : {2, 4, 64}) {
        char nm[96]; snprintf(nm, sizeof nm, "write-only float4, %d blocks/CU", bpc);
        t = time_ms([&] { hipLaunchKernelGGL((k_fill16<false>), dim3(CU * bpc), dim3(256), 0, 0, b16, n16); });
        report(nm, t, 1.0 * n * 32);
      }
    }
    hipFree(a); hipFree(b);
  }
  hipFree(out);
  return 0;
}
