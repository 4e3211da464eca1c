// Cluster build from raw points on gfx950 ("next" row N1 of SURVEY.md 8f): the only stage of the
// path that streams point lists.   reference: include/tools.hpp:311-316 (PointCluster::push),
// src/benchmark/benchmark_virtual.cpp:392-403, src/benchmark/bavoxel.hpp:1194-1198.
// HBM-bound: 20 bytes per point in (3 f32 + two i32 keys), one 80-byte cluster out per (a,i).
// Each wavefront takes 64 consecutive points, reduces runs of equal (feature,pose) keys with a
// segmented shuffle scan (points normally arrive grouped), and issues ten f64 atomics per run.
#include "balm_internal.h"

namespace balm {

__global__ __launch_bounds__(256) void k_build_clusters(const float *__restrict__ xyz, const int *__restrict__ fid,
                                                        const int *__restrict__ pid, long n_pts, int F, int W,
                                                        double *__restrict__ soa) {
  const int lane = threadIdx.x & 63;
  const long nchunk = (n_pts + 63) / 64;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
  for (long ch = wave; ch < nchunk; ch += nwave) {
    const long t = ch * 64 + lane;
    long key = -1;
    double val[10];
#pragma unroll
    for (int c = 0; c < 10; c++) val[c] = 0.0;
    if (t < n_pts) {
      const int a = fid[t], i = pid[t];
      if (a >= 0 && a < F && i >= 0 && i < W) {
        key = (long)a * W + i;
        const double x = xyz[3 * t], y = xyz[3 * t + 1], z = xyz[3 * t + 2];
        val[0] = x * x; val[1] = x * y; val[2] = x * z; val[3] = y * y; val[4] = y * z; val[5] = z * z;
        val[6] = x; val[7] = y; val[8] = z; val[9] = 1.0;
      }
    }
    const long prev = __shfl_up(key, 1, 64);
    const bool head = (lane == 0) || (prev != key);
    const unsigned long long heads = __ballot(head);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      // lanes (lane, lane+off] contain no run head  <=>  lane+off is in my run
      const unsigned long long span = (off == 63 ? ~0ull : ((1ull << off) - 1ull)) << 1;   // bits 1..off
      const bool take = (lane + off < 64) && (((heads >> lane) & span) == 0ull);
#pragma unroll
      for (int c = 0; c < 10; c++) {
        const double nb = __shfl_down(val[c], off, 64);
        if (take) val[c] += nb;
      }
    }
    if (head && key >= 0) {
      const long a = key / W;
      const int i = (int)(key - a * W);
      double *dst = soa + (size_t)a * 10 * W + i;
#pragma unroll
      for (int c = 0; c < 10; c++) atomicAdd(dst + (size_t)c * W, val[c]);
    }
  }
}

void launch_build_clusters(hipStream_t s, const float *xyz, const int *feat_id, const int *pose_id, long n_pts, int F,
                           int W, double *soa) {
  if (n_pts <= 0) return;
  long nchunk = (n_pts + 63) / 64;
  long blocks = (nchunk + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_build_clusters, dim3((unsigned)blocks), dim3(256), 0, s, xyz, feat_id, pose_id, n_pts, F, W, soa);
}

__global__ void k_soa_to_aos(const double *__restrict__ soa, double *__restrict__ aos, int F, int W) {
  const size_t total = (size_t)F * W;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t a = t / W;
    const int i = (int)(t - a * W);
    const double *src = soa + a * 10 * W + i;
    double *dst = aos + t * 10;
#pragma unroll
    for (int c = 0; c < 10; c++) dst[c] = src[(size_t)c * W];
  }
}

void launch_soa_to_aos(hipStream_t s, const double *soa, double *aos, int F, int W) {
  size_t total = (size_t)F * W;
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_soa_to_aos, dim3(grid), dim3(256), 0, s, soa, aos, F, W);
}

}  // namespace balm
