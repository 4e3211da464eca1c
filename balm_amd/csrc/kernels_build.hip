// Cluster build from raw points on gfx950 ("next" row N1 of SURVEY.md 8f): the only stage of the
// path that streams point lists.   reference: include/tools.hpp:311-316 (PointCluster::push),
// src/benchmark/benchmark_virtual.cpp:392-403, src/benchmark/bavoxel.hpp:1194-1198.
// HBM-bound: 20 bytes per point in (3 f32 + two i32 keys), one 80-byte cluster out per (a,i).
//
// k_build_clusters_runs (points grouped by (feature, pose), keys non-decreasing -- the order every driver of the
// reference produces): a wavefront takes 256..512 consecutive points (launch_build_clusters picks the block size), parks their xyz in LDS (coalesced loads, all in flight at once), compacts the
// heads of the runs of equal keys, and then ONE LANE PER RUN pushes the run's points one by one in order with the
// reference's own operation sequence (P += v v^T as a rounded product and a rounded add per entry, v += p, N += 1:
// tools.hpp:311-316 compiled without FMA) -- so a cluster is bit-identical to PointCluster::push, and lanes r, r+1 hold
// the clusters of consecutive poses: ten stores per cluster, 512 contiguous bytes per store instruction.  A run that
// leaves the block is finished by its lane straight from memory, and the wave that owns the next block finds no head
// at its first point: every run has exactly one writer, no atomics.
// The kernel raises a flag if it meets a decreasing or invalid key; the launcher then redoes the build with
// k_build_clusters_atomic, which takes points in any order (segmented shuffle scan per wave + ten f64 atomics per run).
#include <cstdlib>

#include "balm_internal.h"

namespace balm {

// points per wavefront block, parked in LDS: a multiple of 64 (BALM_BUILD_BP = 256 / 384 / 512 selects it for A/B runs)

struct f3 { float x, y, z; };         // 12 bytes, 4-byte aligned: one global_load_dwordx3 per point

template <int BUILD_BP>
__global__ __launch_bounds__(256) void k_build_clusters_runs(const float *__restrict__ xyz, const int *__restrict__ fid,
                                                             const int *__restrict__ pid, long n_pts, int F, int W,
                                                             double *__restrict__ soa, int *__restrict__ unsorted) {
  __shared__ f3 pts[4][BUILD_BP];
  __shared__ short rstart[4][BUILD_BP + 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long nblk = (n_pts + BUILD_BP - 1) / BUILD_BP;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
  const long key_end = (long)F * W;                       // past every valid key: lanes beyond the last point
  const f3 *__restrict__ P3 = reinterpret_cast<const f3 *>(xyz);
  auto key_of = [&](int a, int i) -> long { return (a >= 0 && a < F && i >= 0 && i < W) ? (long)a * W + i : -1; };
  for (long blk = wave; blk < nblk; blk += nwave) {
    const long t0 = blk * BUILD_BP;
    const int nloc = (int)(n_pts - t0 < BUILD_BP ? n_pts - t0 : BUILD_BP);
    // ---- every load of the block is issued before the first use (the kernel is latency-bound otherwise: few waves fit
    // beside 11 KB of LDS each, and a dependent round trip to HBM per 64 points is most of a block's time)
    constexpr int NJ = BUILD_BP / 64;
    f3 q[NJ]; int fa[NJ], fi[NJ];
    long carry = -2;
    if (t0 > 0) carry = key_of(fid[t0 - 1], pid[t0 - 1]);  // key of the point before this block (wave-uniform)
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const long t = t0 + j * 64 + lane;
      q[j] = f3{0.f, 0.f, 0.f}; fa[j] = -1; fi[j] = -1;
      if (t < n_pts) { q[j] = P3[t]; fa[j] = fid[t]; fi[j] = pid[t]; }
    }
    // ---- points and keys into LDS, run heads compacted into rstart: run r = points rstart[r] .. rstart[r+1]-1
    int runs = 0;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int p = j * 64 + lane;
      const long t = t0 + p;
      const long key = t < n_pts ? key_of(fa[j], fi[j]) : key_end;
      pts[wv][p] = q[j];
      long prev = __shfl_up(key, 1, 64);
      if (lane == 0) prev = carry;
      carry = __shfl(key, 63, 64);
      if (t < n_pts && (key < 0 || prev > key)) *unsorted = 1;        // benign race: everyone writes 1
      const bool head = t < n_pts && prev != key;
      const unsigned long long bal = __ballot(head);
      if (head) rstart[wv][runs + __popcll(bal & ((1ull << lane) - 1ull))] = (short)p;
      runs += __popcll(bal);
    }
    if (lane == 0) rstart[wv][runs] = (short)nloc;
    // ---- one lane per run: the reference's push, point by point in order (tools.hpp:311-316, one rounding per operation)
    for (int r = lane; r < runs; r += 64) {
      const int s0 = rstart[wv][r], s1 = rstart[wv][r + 1];
      const int2 ai = make_int2(fid[t0 + s0], pid[t0 + s0]);       // read a moment ago by this wave: a cache hit
      const long key = key_of(ai.x, ai.y);
      double pxx = 0, pxy = 0, pxz = 0, pyy = 0, pyz = 0, pzz = 0, vx = 0, vy = 0, vz = 0, cnt = 0;
      auto push = [&](const f3 v) {
        const double dx = v.x, dy = v.y, dz = v.z;
        pxx = __dadd_rn(pxx, __dmul_rn(dx, dx)); pxy = __dadd_rn(pxy, __dmul_rn(dx, dy)); pxz = __dadd_rn(pxz, __dmul_rn(dx, dz));
        pyy = __dadd_rn(pyy, __dmul_rn(dy, dy)); pyz = __dadd_rn(pyz, __dmul_rn(dy, dz)); pzz = __dadd_rn(pzz, __dmul_rn(dz, dz));
        vx = __dadd_rn(vx, dx); vy = __dadd_rn(vy, dy); vz = __dadd_rn(vz, dz);
        cnt += 1.0;
      };
      for (int l = s0; l < s1; l++) push(pts[wv][l]);
      if (s1 == nloc)      // the block's last run may go on in the next blocks: its lane finishes it from memory,
        for (long tt = t0 + nloc; tt < n_pts && key_of(fid[tt], pid[tt]) == key; tt++) push(P3[tt]);
      if (key >= 0) {      // ... and the wave that owns the next block finds no head at its first point
        double *dst = soa + (size_t)ai.x * 10 * W + ai.y;
        dst[0] = pxx; dst[(size_t)W] = pxy; dst[(size_t)2 * W] = pxz; dst[(size_t)3 * W] = pyy; dst[(size_t)4 * W] = pyz;
        dst[(size_t)5 * W] = pzz; dst[(size_t)6 * W] = vx; dst[(size_t)7 * W] = vy; dst[(size_t)8 * W] = vz; dst[(size_t)9 * W] = cnt;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_build_clusters_atomic(const float *__restrict__ xyz, const int *__restrict__ fid,
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

// `flag` = one device int, zero on entry; returns after enqueueing.  The caller reads it back: non-zero = the points were
// not grouped, re-zero the table and call launch_build_clusters_any.
void launch_build_clusters(hipStream_t s, const float *xyz, const int *feat_id, const int *pose_id, long n_pts, int F,
                           int W, double *soa, int *flag) {
  if (n_pts <= 0) return;
  // points per wavefront block: a block's runs are pushed by one lane each, 64 at a time, so the block size that wastes
  // the fewest lanes is the one that holds a whole number of 64-run rounds.  With l = points per run (the average over the
  // table: n_pts / (F W)) a block of 64 l points is exactly one round -- 6-point runs: 384 points, 0.169 ms for 24 M
  // points instead of 0.205 with 512 (85 runs = a full round and a third of one); long runs take the largest block.
  const char *e = getenv("BALM_BUILD_BP");               // A/B runs: 256 / 320 / 384 / 448 / 512
  int bp = e ? atoi(e) : 0;
  if (!bp) {
    const double len = (double)n_pts / ((double)F * W > 1 ? (double)F * W : 1.0);
    double best = -1;
    for (int cand = 256; cand <= 512; cand += 64) {
      const double runs = cand / (len < 1 ? 1.0 : len);
      const double rounds = runs <= 64 ? 1.0 : (double)(long)((runs + 63.999) / 64);
      const double eff = runs / (64.0 * rounds);
      if (eff >= best - 1e-9) { best = eff; bp = cand; }
    }
  }
  long nblk = (n_pts + bp - 1) / bp;
  long blocks = (nblk + 3) / 4;
  if (blocks > 16384) blocks = 16384;
#define BALM_LAUNCH_RUNS(BP)                                                                                              \
  hipLaunchKernelGGL(k_build_clusters_runs<BP>, dim3((unsigned)blocks), dim3(256), 0, s, xyz, feat_id, pose_id, n_pts, F, W, soa, flag)
  switch (bp) {
    case 256: BALM_LAUNCH_RUNS(256); break;
    case 320: BALM_LAUNCH_RUNS(320); break;
    case 384: BALM_LAUNCH_RUNS(384); break;
    case 448: BALM_LAUNCH_RUNS(448); break;
    default: BALM_LAUNCH_RUNS(512); break;
  }
#undef BALM_LAUNCH_RUNS
}

void launch_build_clusters_any(hipStream_t s, const float *xyz, const int *feat_id, const int *pose_id, long n_pts, int F,
                               int W, double *soa) {
  if (n_pts <= 0) return;
  long nchunk = (n_pts + 63) / 64;
  long blocks = (nchunk + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_build_clusters_atomic, dim3((unsigned)blocks), dim3(256), 0, s, xyz, feat_id, pose_id, n_pts, F, W, soa);
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
