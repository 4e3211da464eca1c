// Adaptive-voxel point association on gfx950 ("next" row N3 of SURVEY.md 8f): the last CPU stage of
// the real-world pipeline.  Same decisions as the host restatement the tests compare it with, i.e. as the
// reference's cut_voxel / recut / tras_opt (src/benchmark/bavoxel.hpp:1170-1223, :654-776, :908-929):
// a point's root voxel, octants and every plane test are evaluated with the reference's float/double
// types and WITHOUT fused multiply-adds (this file is compiled -ffp-contract=off; the reference is built
// without -march, so it has none), so integer/index results are bit-exact.
//
// Instead of growing an octree point by point the GPU evaluates all three levels at once:
//   1  per point: world position, root voxel key, octant at level 1 and level 2
//   2  radix sort by root key -> dense root ids; then, per level L = 0,1,2, a stable radix sort by
//      (root, octant prefix of L, frame): the scan order of the points survives inside every
//      (node, frame) segment, which is the order the reference pushes them in (cut_voxel / cut_func)
//   3  one lane per segment accumulates the body-frame and world-frame PointCluster sequentially
//      (tools.hpp:311-316 order -> the sums are bit-exact at every level); node totals add the
//      per-frame world clusters in frame order (judge_eigen's loop)
//   4  plane tests top-down (recut), feature list (tras_opt + push_voxel's >= 2 observing poses),
//      per-(feature, pose) body clusters copied from the feature's level.
// HBM-bound: ~20 B/point read a handful of times; the four sorts dominate.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "balm_internal.h"

namespace balm {

namespace {

struct VoxParams {
  double voxel_size;
  float thr[3];
  int min_ps;
  int W;                  // scans, the marginalised ones included
  int layer_limit;        // deepest octree layer that may become a feature (bavoxel.hpp:8; 0..2)
  int min_observers;      // push_voxel's minimum number of observing scans (bavoxel.hpp:32-37: 2; BAs_left.hpp:38: none)
  int fix_frames;         // leading scans folded into world-frame fix clusters (to_margi, bavoxel.hpp:778-816)
  double max_dis, ratio21_max, lam0_max;     // the consistency driver's plane test (BAs_left.hpp:674); 0 = off
};

__device__ __forceinline__ void world_point(const float *__restrict__ xyz, const double *__restrict__ pose, long p,
                                            double q[3], double po[3]) {
  po[0] = (double)xyz[3 * p]; po[1] = (double)xyz[3 * p + 1]; po[2] = (double)xyz[3 * p + 2];
#pragma unroll
  for (int r = 0; r < 3; r++)   // ((R(r,0) x + R(r,1) y) + R(r,2) z) + t(r), one rounding per operation
    q[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(pose[r], po[0]), __dmul_rn(pose[3 + r], po[1])),
                               __dmul_rn(pose[6 + r], po[2])), pose[9 + r]);
}

// The scan a point belongs to: the caller's per-point array (balm_associate), or -- balm_associate_scans, where the scans arrive as
// containers -- the offsets of the scans in the packed point list: first[k] <= p < first[k + 1] (m scans, empty ones allowed).  The
// per-point array then never exists: 4 bytes per point not uploaded, not expanded (53 us for the shipped window), not read by passes A, B.
struct ScanOf {
  const int *frame;
  const long *first;
  int m;
  __device__ __forceinline__ int find(long p, const long *tbl) const {           // p < first[m]; tbl = first, or its LDS copy
    if (frame) return frame[p];
    int lo = 0, hi = m;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tbl[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  }
  // the scan of p, given the scan `fr` of an earlier point (a kernel's points ascend: almost always no step)
  __device__ __forceinline__ int advance(int fr, long p, const long *tbl) const {
    if (frame) return frame[p];
    while (fr + 1 < m && p >= tbl[fr + 1]) fr++;
    return fr;
  }
};

// cut_voxel's key (bavoxel.hpp:1178-1184): loc = (float)(q / voxel_size), shifted down for negatives, truncated
// (Measured, round 5: q * (1 / voxel_size) where the voxel size is a power of two -- the same correctly rounded number without the
// three FP64 divisions per point -- and the pose through the scalar cache: passes A and B of balm_associate did not move, 94 / 83 us;
// they are bound by their 12-byte-stride point loads, not by instructions.  profiles/r05w_head_scan.txt)
__device__ __forceinline__ long long voxel_key(double q, double vs) {
  float loc = (float)(q / vs);
  if (loc < 0) loc = (float)((double)loc - 1.0);
  return (long long)loc;
}

// pass A: range of the voxel keys per axis (to pack them into as few radix digits as possible); keys are
// saturated to +-2^30 here -- anything beyond 2^21 voxels per axis is rejected by the host anyway.
// One row of {min[3], max[3], bad} per block; the host folds the rows.  bad != 0: a scan index outside [0, W) or a
// non-finite coordinate (input validation rides along instead of a host pass over the points).
constexpr int RANGE_BLOCKS = 2048;
constexpr int RANGE_ROW = 7;
__global__ __launch_bounds__(256) void k_vox_range(const float *__restrict__ xyz, const ScanOf scan,
                                                   const double *__restrict__ poses, long n, int W, double vs,
                                                   int *__restrict__ range /* [gridDim.x][RANGE_ROW] */) {
  const int *__restrict__ frame = scan.frame;
  __shared__ int red[4][RANGE_ROW];
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
  int bad = 0;
  // RANGE_U points per trip, their loads issued together, on up to 2 048 blocks (a full complement of resident wavefronts): with one point
  // per trip on 1 024 blocks the pass had 5 MB in flight and ran at 2.3 TB/s -- latency-bound (94 us for the shipped window's 215 MB).
  // A workgroup walks ONE contiguous stretch of the points (round 6): with the scans given by their offsets the scan of the stretch's first
  // point is searched once and the next four scan boundaries sit in LDS -- a point's scan is four compares, no dependent load.
  constexpr int RANGE_U = 4;
  const long per = (n + gridDim.x - 1) / gridDim.x, beg = (long)blockIdx.x * per, end = min(n, beg + per);
  __shared__ int s_f0;
  __shared__ long s_b[4];
  if (!frame) {
    if (threadIdx.x == 0 && beg < end) {
      const int f0 = scan.find(beg, scan.first);
      s_f0 = f0;
      for (int j = 0; j < 4; j++) s_b[j] = scan.first[min(f0 + 1 + j, scan.m)];
    }
    __syncthreads();
  }
  for (long p0 = beg + threadIdx.x; p0 < end; p0 += RANGE_U * (long)blockDim.x) {
    long pp[RANGE_U];
    int frv[RANGE_U], frp[RANGE_U];
    float xv[RANGE_U][3];
#pragma unroll
    for (int u = 0; u < RANGE_U; u++) pp[u] = p0 + u * (long)blockDim.x;
#pragma unroll
    for (int u = 0; u < RANGE_U; u++) {
      const bool in = pp[u] < end;
      const long p = in ? pp[u] : p0;
      if (frame) { frv[u] = frame[p]; frp[u] = p > 0 ? frame[p - 1] : frv[u]; }
      xv[u][0] = xyz[3 * p]; xv[u][1] = xyz[3 * p + 1]; xv[u][2] = xyz[3 * p + 2];
    }
    if (!frame) {
#pragma unroll
      for (int u = 0; u < RANGE_U; u++) {
        const long p = pp[u] < end ? pp[u] : p0;
        int fr = s_f0 + (p >= s_b[0] ? 1 : 0) + (p >= s_b[1] ? 1 : 0) + (p >= s_b[2] ? 1 : 0) + (p >= s_b[3] ? 1 : 0);
        if (p >= s_b[3]) fr = scan.advance(fr, p, scan.first);      // (empty or tiny scans: walk on)
        frv[u] = frp[u] = min(fr, scan.m - 1);
      }
    }
#pragma unroll
    for (int u = 0; u < RANGE_U; u++) {
      if (pp[u] >= end) continue;
      const int fr = frv[u];
      if (frp[u] > fr) bad |= 2;                              // not in scan order: the level sorts must sort the scan bits too
      if (fr < 0 || fr >= W) { bad |= 1; continue; }
      const double *pose = poses + 12 * (long)fr;
      const double po[3] = {(double)xv[u][0], (double)xv[u][1], (double)xv[u][2]};
      double q[3];
#pragma unroll
      for (int r = 0; r < 3; r++)   // world_point's arithmetic: ((R(r,0) x + R(r,1) y) + R(r,2) z) + t(r), one rounding per operation
        q[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(pose[r], po[0]), __dmul_rn(pose[3 + r], po[1])),
                                   __dmul_rn(pose[6 + r], po[2])), pose[9 + r]);
      if (!(isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]))) { bad |= 1; continue; }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const long long k = voxel_key(q[j], vs);
        const int ki = (int)max(-(1ll << 30), min(1ll << 30, k));
        lo[j] = min(lo[j], ki); hi[j] = max(hi[j], ki);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++)
    for (int d = 32; d >= 1; d >>= 1) {
      lo[j] = min(lo[j], __shfl_xor(lo[j], d, 64));
      hi[j] = max(hi[j], __shfl_xor(hi[j], d, 64));
    }
  for (int d = 32; d >= 1; d >>= 1) bad |= __shfl_xor(bad, d, 64);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 3; j++) { red[threadIdx.x >> 6][j] = lo[j]; red[threadIdx.x >> 6][3 + j] = hi[j]; }
    red[threadIdx.x >> 6][6] = bad;
  }
  __syncthreads();
  if (threadIdx.x < RANGE_ROW) {
    int v = red[0][threadIdx.x];
    for (int w = 1; w < 4; w++)
      v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : (threadIdx.x == 6 ? (v | red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]));
    range[blockIdx.x * RANGE_ROW + threadIdx.x] = v;
  }
}

// root key = mixed-radix number of the three per-axis voxel indices: ((kx - off_x) * n_y + (ky - off_y)) * n_z + (kz - off_z), n = cells
// per axis.  Any injective map does (the sort only has to bring equal keys together); this one needs ceil(log2(n_x n_y n_z)) bits
// instead of the sum of the axes' bit counts -- 24 instead of 25 on the shipped window (223 x 389 x 178 cells): three 8-bit radix
// passes instead of four.
struct KeyPack { long long off[3]; unsigned long long n[3]; };

// pass B: packed root key + cut_func's octants (bavoxel.hpp:709-720) for both subdivision levels; the sort
// value carries (point index, octants, frame) so that later passes never gather per-point attributes
template <class K>
__global__ __launch_bounds__(256) void k_vox_keys(const float *__restrict__ xyz, const ScanOf scan,
                                                  const double *__restrict__ poses, long n, double vs, KeyPack kp,
                                                  K *__restrict__ k0, unsigned long long *__restrict__ val) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double q[3], po[3];
  const int fr = scan.find(p, scan.first);
  world_point(xyz, poses + 12 * (long)fr, p, q, po);
  const float q1 = (float)(vs / 4.0);
  unsigned long long key = 0;
  int o1 = 0, o2 = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const long long kj = voxel_key(q[j], vs);
    const float c0 = (float)((0.5 + (double)kj) * vs);
    const int b1 = q[j] > (double)c0;
    const float c1 = __fadd_rn(c0, __fmul_rn((float)(2 * b1 - 1), q1));
    const int b2 = q[j] > (double)c1;
    key = key * kp.n[j] + (unsigned long long)(kj - kp.off[j]);
    o1 = (o1 << 1) | b1;
    o2 = (o2 << 1) | b2;
  }
  k0[p] = (K)key;
  val[p] = ((unsigned long long)p << 15) | ((unsigned long long)((o1 << 3) | o2) << 9) | (unsigned long long)fr;
}

template <class T>
__global__ void k_head_flags(const T *__restrict__ key, long n, int shift, unsigned int *__restrict__ flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || (key[i] >> shift) != (key[i - 1] >> shift)) ? 1u : 0u;
}

// composite key of level L: (root id, octant prefix of 3 L bits, frame of fb bits) -- as few radix digits as the
// window needs; the canonical form (root << 15 | octants << 9 | frame) is restored per segment in k_seg_heads
template <class K>
__global__ void k_make_ck(const unsigned int *__restrict__ rootid_incl, const unsigned long long *__restrict__ val, long n,
                          int level, int fb, K *__restrict__ ck, unsigned int *__restrict__ idx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long v = val[i];
  const unsigned long long oct = ((v >> 9) & 0x3full) >> (6 - 3 * level);       // o1 for level 1, (o1, o2) for level 2
  ck[i] = (K)(((((unsigned long long)(rootid_incl[i] - 1) << (3 * level)) | oct) << fb) | (v & 0x1ffull));
  idx[i] = (unsigned int)(v >> 15);
}

// segment s = run of equal composite keys: its first position and its key in canonical form
// (the entry behind the last segment's -- seg_start[NS] = n -- and the zeroed counters of the segment kernels' work lists ride along:
//  one launch and one fill node less per level)
template <class K>
__global__ void k_seg_heads(const K *__restrict__ cks, const unsigned int *__restrict__ segid_incl, long n, int level, int fb,
                            unsigned int *__restrict__ seg_start, unsigned long long *__restrict__ seg_ck,
                            unsigned int *__restrict__ counters4 = nullptr) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) seg_start[segid_incl[i]] = (unsigned int)n;
  if (i < 4 && counters4) counters4[i] = 0u;
  if (i == 0 || cks[i] != cks[i - 1]) {
    const unsigned int s = segid_incl[i] - 1;
    const unsigned long long k = cks[i];
    const unsigned long long frame = k & ((1ull << fb) - 1), rest = k >> fb;
    const unsigned long long oct = (rest & ((1ull << (3 * level)) - 1)) << (6 - 3 * level), root = rest >> (3 * level);
    seg_start[s] = (unsigned int)i;
    seg_ck[s] = (root << 15) | (oct << 9) | frame;
  }
}

// PointCluster::push in scan order (tools.hpp:311-316), body frame (sig_orig) and world frame (sig_tran).
// The sums are sequential in scan order (that is what makes them equal to the reference's bit for bit), so
// the parallelism is across segments and across the 18 sums of a segment:
//   short segments (<= SEG_SHORT points): one lane per segment
//   long segments: one wave per segment; 64 points at a time are gathered and expanded into their 18 terms in
//   parallel, parked in LDS, and lanes 0..17 each add one term column in order.
constexpr int SEG_SHORT = 8;           // (24 until round 3: a lane's gathers are now issued as one batch, which has to fit its registers)
constexpr int SEG_TERMS = 18;          // 6 + 3 body, 6 + 3 world; N is the segment length
// padded LDS row of a term column.  66, not 65: an adding lane reads 16 bytes per instruction (ds_read2_b64) from its own column, i.e.
// lane j from bank 2 j LD mod 64 on -- with 65 two neighbouring lanes share two banks in every read (a 2-way conflict on the
// instruction that bounds k_seg_wave), with 66 sixteen lanes take sixteen disjoint bank quads.  Shipped window, association on the
// device: 2.74-2.82 ms with 65, 2.66 with 66, 2.72 with 67, 3.11 with 64 (profiles/r05u_segment_lds_row.txt).
constexpr int SEG_LD = 66;

struct PointTerms { double t[SEG_TERMS]; };

__device__ __forceinline__ PointTerms point_terms(const float x[3], const double P[12]) {
  const double po[3] = {(double)x[0], (double)x[1], (double)x[2]};
  double q[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
    q[r] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(P[r], po[0]), __dmul_rn(P[3 + r], po[1])), __dmul_rn(P[6 + r], po[2])), P[9 + r]);
  PointTerms o;
  o.t[0] = __dmul_rn(po[0], po[0]); o.t[1] = __dmul_rn(po[0], po[1]); o.t[2] = __dmul_rn(po[0], po[2]);
  o.t[3] = __dmul_rn(po[1], po[1]); o.t[4] = __dmul_rn(po[1], po[2]); o.t[5] = __dmul_rn(po[2], po[2]);
  o.t[6] = po[0]; o.t[7] = po[1]; o.t[8] = po[2];
  o.t[9] = __dmul_rn(q[0], q[0]); o.t[10] = __dmul_rn(q[0], q[1]); o.t[11] = __dmul_rn(q[0], q[2]);
  o.t[12] = __dmul_rn(q[1], q[1]); o.t[13] = __dmul_rn(q[1], q[2]); o.t[14] = __dmul_rn(q[2], q[2]);
  o.t[15] = q[0]; o.t[16] = q[1]; o.t[17] = q[2];
  return o;
}

// ONE launch for both classes (round 3): blocks [0, short_blocks) give a lane to every segment and skip the long ones, the
// blocks behind them give a wavefront to every segment and skip the short ones -- the two classes used to be two launches
// that ran one after the other; now the long pole (one wavefront walking the longest segment) hides the short ones.
// ns_dev != NULL: the segment count is still on the device (the window map's recut) and the grid covers a bound.
__global__ __launch_bounds__(256) void k_seg_clusters(const float *__restrict__ xyz, const double *__restrict__ poses,
                                                      const unsigned int *__restrict__ idx, const unsigned int *__restrict__ seg_start,
                                                      const unsigned long long *__restrict__ seg_ck, long NS, int short_blocks,
                                                      double *__restrict__ seg_body, double *__restrict__ seg_world,
                                                      const unsigned int *__restrict__ ns_dev) {
  __shared__ double lds[4][SEG_TERMS * SEG_LD];
  if (ns_dev) NS = *ns_dev;
  if ((int)blockIdx.x < short_blocks) {
    // ---- short segments: one lane each.  The gathers (index -> coordinates) of up to SEG_SHORT points are issued together:
    // a lane that walked them one by one paid two dependent memory round trips per point (12.7 us per launch on a scan's
    // 2 000 segments, most of it waiting).
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    const unsigned int i0 = seg_start[s], i1 = seg_start[s + 1];
    const int cnt = (int)(i1 - i0);
    if (cnt > SEG_SHORT) return;
    const double *pose = poses + 12 * (long)(seg_ck[s] & 511ull);
    double P[12];
#pragma unroll
    for (int c = 0; c < 12; c++) P[c] = pose[c];
    double acc[SEG_TERMS];
#pragma unroll
    for (int c = 0; c < SEG_TERMS; c++) acc[c] = 0.0;
    size_t pp[SEG_SHORT];
#pragma unroll
    for (int u = 0; u < SEG_SHORT; u++) pp[u] = idx[i0 + (u < cnt ? u : 0)];
    float xx[SEG_SHORT][3];
#pragma unroll
    for (int u = 0; u < SEG_SHORT; u++) { xx[u][0] = xyz[3 * pp[u]]; xx[u][1] = xyz[3 * pp[u] + 1]; xx[u][2] = xyz[3 * pp[u] + 2]; }
#pragma unroll
    for (int u = 0; u < SEG_SHORT; u++) {
      if (u < cnt) {
        const PointTerms t = point_terms(xx[u], P);
#pragma unroll
        for (int c = 0; c < SEG_TERMS; c++) acc[c] = __dadd_rn(acc[c], t.t[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 9; c++) { seg_body[s * 10 + c] = acc[c]; seg_world[s * 10 + c] = acc[9 + c]; }
    seg_body[s * 10 + 9] = seg_world[s * 10 + 9] = (double)cnt;
    return;
  }
  // ---- long segments: one wavefront each; 64 points at a time are gathered and expanded into their 18 terms in parallel,
  // parked in LDS, and lanes 0..17 each add one term column in order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long s = ((long)blockIdx.x - short_blocks) * 4 + wv;
  if (s >= NS) return;
  const unsigned int i0 = seg_start[s], i1 = seg_start[s + 1];
  if (i1 - i0 <= SEG_SHORT) return;
  const double *pose = poses + 12 * (long)(seg_ck[s] & 511ull);
  double P[12];
#pragma unroll
  for (int c = 0; c < 12; c++) P[c] = pose[c];
  double *mine = lds[wv];
  const int col = min(lane, SEG_TERMS - 1);
  double acc = 0.0;
  // two-deep software pipeline over the dependent gathers: index of chunk k+2, coordinates of chunk k+1
  size_t pn = idx[min(i0 + lane, i1 - 1)];
  float xn[3] = {xyz[3 * pn], xyz[3 * pn + 1], xyz[3 * pn + 2]};
  pn = idx[min(i0 + 64 + lane, i1 - 1)];
  for (unsigned int i = i0; i < i1; i += 64) {
    const int cnt = (int)min(64u, i1 - i);
    const float x[3] = {xn[0], xn[1], xn[2]};
    xn[0] = xyz[3 * pn]; xn[1] = xyz[3 * pn + 1]; xn[2] = xyz[3 * pn + 2];
    pn = idx[min(i + 128 + lane, i1 - 1)];
    const PointTerms t = point_terms(x, P);
    // the chunk padded with +0.0 (x + 0.0 = x bit for bit for every x these sums can hold) and ONE fully unrolled chain of 64 additions
    // whose LDS reads run sixteen values ahead -- k_seg_wave's add phase (round 5).  The rolled loop with its tail paid an LDS round trip
    // per eight additions, ~1.5 us per chunk: the longest segment of a scan set the launch's length (23.6 us, 2.75 times per
    // balm_window_add_scan).
    const bool in = lane < cnt;
#pragma unroll
    for (int c = 0; c < SEG_TERMS; c++) mine[c * SEG_LD + lane] = in ? t.t[c] : 0.0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the wave's LDS writes have landed
    if (lane < SEG_TERMS) {
      const double *colp = mine + col * SEG_LD;
      double va[16], vb[16];
#pragma unroll
      for (int u = 0; u < 16; u++) va[u] = colp[u];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        if (b < 3) {
#pragma unroll
          for (int u = 0; u < 16; u++) vb[u] = colp[16 * (b + 1) + u];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __dadd_rn(acc, va[u]);
#pragma unroll
        for (int u = 0; u < 16; u++) va[u] = vb[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 9) seg_body[s * 10 + lane] = acc;
  else if (lane < SEG_TERMS) seg_world[s * 10 + lane - 9] = acc;
  else if (lane == SEG_TERMS) seg_body[s * 10 + 9] = seg_world[s * 10 + 9] = (double)(i1 - i0);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the batch association without the library sorts inside root voxels, and without gathers.
//   * the root sort moves (key, point index) only; ONE gather then lays the points down in root order as 16-byte records
//     {x, y, z, (o1, o2, scan)} -- every later pass streams them (the three segment kernels used to gather 12 bytes per point out
//     of 64-byte lines through the sorted indices: 350 us per level on the shipped window, 1.05 ms of the association's 3.6);
//   * levels 1 and 2 are a STABLE PARTITION inside each root voxel by the 3 / 6 octant bits (the list is already grouped by root and in
//     scan order inside it): per-tile histograms, per-root offsets, one scatter that writes both levels -- one read of the records
//     instead of the library's make-key + 2 + 3 onesweep passes (9 launches of ~71 us + 2 x 44).  Stable, so the points of a (node, scan)
//     segment stay in scan order: the sums below are the reference's, bit for bit.
// Points that do not arrive scan by scan, or windows whose level-2 key needs more than 32 bits, take the sorted path above.
constexpr int PART_TS = 2048;          // records per partition tile (a tile never spans two roots)

// pass B of the fast path: root key + the record's tag (o1 << 12 | o2 << 9 | scan); the sort value is the point's index
template <class K>
__global__ __launch_bounds__(256) void k_vox_keys_tag(const float *__restrict__ xyz, const ScanOf scan,
                                                      const double *__restrict__ poses, long n, double vs, KeyPack kp,
                                                      K *__restrict__ k0, unsigned int *__restrict__ idx, unsigned short *__restrict__ tag) {
  // offsets: the scan of the workgroup's first point is found once (one thread, nine dependent loads that hit the cache: every
  // workgroup reads the same 1.4 KB); the others step on from it -- a point's scan is almost always its workgroup's first point's
  __shared__ int s_f0;
  __shared__ long s_b[2];        // where the next two scans begin: a workgroup's 256 points rarely reach past them
  if (!scan.frame) {
    if (threadIdx.x == 0) {
      const int f0 = scan.find(min((long)blockIdx.x * blockDim.x, n - 1), scan.first);
      s_f0 = f0;
      s_b[0] = scan.first[min(f0 + 1, scan.m)]; s_b[1] = scan.first[min(f0 + 2, scan.m)];
    }
    __syncthreads();
  }
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double q[3], po[3];
  int fr;
  if (scan.frame) fr = scan.frame[p];
  else {
    fr = s_f0 + (p >= s_b[0] ? 1 : 0) + (p >= s_b[1] ? 1 : 0);
    if (p >= s_b[1]) fr = scan.advance(fr, p, scan.first);       // (empty or tiny scans: walk on)
    fr = min(fr, scan.m - 1);
  }
  world_point(xyz, poses + 12 * (long)fr, p, q, po);
  const float q1 = (float)(vs / 4.0);
  unsigned long long key = 0;
  int o1 = 0, o2 = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const long long kj = voxel_key(q[j], vs);
    const float c0 = (float)((0.5 + (double)kj) * vs);
    const int b1 = q[j] > (double)c0;
    const float c1 = __fadd_rn(c0, __fmul_rn((float)(2 * b1 - 1), q1));
    const int b2 = q[j] > (double)c1;
    key = key * kp.n[j] + (unsigned long long)(kj - kp.off[j]);
    o1 = (o1 << 1) | b1;
    o2 = (o2 << 1) | b2;
  }
  k0[p] = (K)key;
  idx[p] = (unsigned int)p;
  tag[p] = (unsigned short)((o1 << 12) | (o2 << 9) | fr);
}

// the points in root order: rec[i] = {x, y, z bits, tag}; level-0 composite key (root, scan); the tags again as a dense 2-byte stream
// for the histogram pass
__global__ __launch_bounds__(256) void k_gather_records(const float *__restrict__ xyz, const unsigned short *__restrict__ tag,
                                                        const unsigned int *__restrict__ idxs, const unsigned int *__restrict__ rootid_incl,
                                                        long n, int fb, uint4 *__restrict__ rec, unsigned int *__restrict__ ck0,
                                                        unsigned short *__restrict__ tags) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t p = idxs[i];
  const unsigned int t = tag[p];
  rec[i] = make_uint4(__float_as_uint(xyz[3 * p]), __float_as_uint(xyz[3 * p + 1]), __float_as_uint(xyz[3 * p + 2]), t);
  ck0[i] = ((rootid_incl[i] - 1) << fb) | (t & 511u);
  tags[i] = (unsigned short)t;
}

// ---- round 6: the root order by a stable MULTISPLIT instead of a radix sort ------------------------------------------------------------
// The packed root key has 24 bits on the shipped window -- and 1 633 VALUES: the points of a lidar window fall into a few thousand root
// voxels, metres wide.  Sorting 13.4 M (key, index) pairs through three 8-bit radix passes, ranking the sorted keys and then gathering the
// points by sorted index (5 x 74 + 30 + 254 + 21 us) does far more than the order needs:
//   M1  k_ms_keys     pass B as before (key, tag) + the key's bit in a bitmap of the key space (2^key_bits bits: 2 MB)
//   M2  rank          exclusive prefix popcount over the bitmap words: a key's DENSE root index = words before + bits below, in key order
//                     -- the numbering the sort + head-flag scan produced
//   M3  k_ms_hist     per tile of 8 192 consecutive points: how many fall into each root (LDS histogram); the dense index is kept per point
//   M4  k_ms_group_sums / k_ms_bases / k_ms_tile_offsets   per root, an exclusive scan over the tiles (in 32 groups) on top of the roots' own
//                     exclusive scan: where each tile's points of each root go.  Also root_start, which k_root_starts used to find.
//   M5  k_ms_scatter  every point's record {x, y, z, tag} straight to its place: rank among the tile's points of its root = counts of
//                     the earlier rounds + of the earlier wavefronts of this round + the lower lanes with the same root (ballots over the
//                     index bits) -- stable, so a root's points stay in scan order and every sum downstream is the reference's, bit for bit.
// ~60 bytes per point in two streaming passes instead of ~150 with a random gather.  Used when the fast path is certain and the window has at
// most MS_MAX_ROOTS root voxels (else the radix path, unchanged); BALM_ASSOC=radix forces the radix path (A/B, tests).
constexpr int MS_TILE = 8192, MS_MAX_ROOTS = 2048, MS_GROUPS = 32;

template <class K>
__global__ __launch_bounds__(256) void k_ms_keys(const float *__restrict__ xyz, const ScanOf scan, const double *__restrict__ poses, long n,
                                                 double vs, KeyPack kp, K *__restrict__ k0, unsigned short *__restrict__ tag,
                                                 unsigned int *__restrict__ bitmap) {
  __shared__ int s_f0;
  __shared__ long s_b[2];
  __shared__ unsigned int s_set[512];
  s_set[threadIdx.x] = 0xffffffffu; s_set[threadIdx.x + 256] = 0xffffffffu;
  if (!scan.frame && threadIdx.x == 0) {
    const int f0 = scan.find(min((long)blockIdx.x * blockDim.x, n - 1), scan.first);
    s_f0 = f0;
    s_b[0] = scan.first[min(f0 + 1, scan.m)]; s_b[1] = scan.first[min(f0 + 2, scan.m)];
  }
  __syncthreads();
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  double q[3], po[3];
  int fr;
  if (scan.frame) fr = scan.frame[p];
  else {
    fr = s_f0 + (p >= s_b[0] ? 1 : 0) + (p >= s_b[1] ? 1 : 0);
    if (p >= s_b[1]) fr = scan.advance(fr, p, scan.first);
    fr = min(fr, scan.m - 1);
  }
  world_point(xyz, poses + 12 * (long)fr, p, q, po);
  const float q1 = (float)(vs / 4.0);
  unsigned long long key = 0;
  int o1 = 0, o2 = 0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const long long kj = voxel_key(q[j], vs);
    const float c0 = (float)((0.5 + (double)kj) * vs);
    const int b1 = q[j] > (double)c0;
    const float c1 = __fadd_rn(c0, __fmul_rn((float)(2 * b1 - 1), q1));
    const int b2 = q[j] > (double)c1;
    key = key * kp.n[j] + (unsigned long long)(kj - kp.off[j]);
    o1 = (o1 << 1) | b1;
    o2 = (o2 << 1) | b2;
  }
  k0[p] = (K)key;
  tag[p] = (unsigned short)((o1 << 12) | (o2 << 9) | fr);
  // The key's bit.  Device-wide looks at a few thousand hot words are expensive on eight XCDs (coherent accesses go past the L2s: one look
  // per run of equal keys cost 290 us on the shipped window), so the workgroup thins them out first: the first lane of every run of equal
  // keys claims the key in a small LDS set; only the one that claims it for the workgroup goes to memory -- a dozen per 256 points.
  const unsigned int kk = (unsigned int)key, prev = __shfl_up(kk, 1, 64);
  if ((threadIdx.x & 63) == 0 || prev != kk) {
    unsigned int slot = (kk * 2654435761u) >> 23;           // 512 slots
    for (int tries = 0; tries < 512; tries++) {
      const unsigned int old = atomicCAS(&s_set[slot], 0xffffffffu, kk);
      if (old == kk) break;                                   // somebody of this workgroup has it
      if (old == 0xffffffffu) {
        const unsigned int w = kk >> 5, bit = 1u << (kk & 31u);
        if (!(__hip_atomic_load(bitmap + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(bitmap + w, bit);
        break;
      }
      slot = (slot + 1) & 511u;
    }
  }
}

struct PopcOf {
  const unsigned int *w; long nw;
  __device__ unsigned int operator()(long i) const { return i < nw ? (unsigned int)__popc(w[i]) : 0u; }
};

__device__ __forceinline__ unsigned int ms_root_of(unsigned int key, const unsigned int *__restrict__ bitmap, const unsigned int *__restrict__ rank_base) {
  const unsigned int w = key >> 5;
  return rank_base[w] + (unsigned int)__popc(bitmap[w] & ((1u << (key & 31u)) - 1u));
}

// hist[tile][r] = points of tile `tile` in root r; rid[p] = the point's dense root index
__global__ __launch_bounds__(256) void k_ms_hist(const unsigned int *__restrict__ key, long n, const unsigned int *__restrict__ bitmap,
                                                 const unsigned int *__restrict__ rank_base, int NR, unsigned short *__restrict__ rid,
                                                 unsigned int *__restrict__ hist) {
  __shared__ unsigned int h[MS_MAX_ROOTS];
  for (int t = threadIdx.x; t < NR; t += 256) h[t] = 0;
  __syncthreads();
  const long a = (long)blockIdx.x * MS_TILE, e = min(n, a + MS_TILE);
  for (long p = a + threadIdx.x; p < e; p += 256) {
    const unsigned int r = ms_root_of(key[p], bitmap, rank_base);
    rid[p] = (unsigned short)r;
    // (runs of equal roots inside a wavefront: one LDS atomic per run)
    const unsigned int prev = __shfl_up(r, 1, 64);
    const bool head = (threadIdx.x & 63) == 0 || prev != r;
    const unsigned long long heads = __ballot(head), act = __ballot(true);      // (both by every lane that still has a point)
    if (head) {
      const int lane = threadIdx.x & 63;
      const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int next = above ? lane + 1 + __builtin_ctzll(above) : 64 - __builtin_clzll(act);
      atomicAdd(&h[r], (unsigned int)(next - lane));
    }
  }
  __syncthreads();
  unsigned int *row = hist + (size_t)blockIdx.x * NR;
  for (int t = threadIdx.x; t < NR; t += 256) row[t] = h[t];
}

// gsum[g][r] = points of root r in the tiles of group g
__global__ __launch_bounds__(256) void k_ms_group_sums(const unsigned int *__restrict__ hist, int NR, long ntiles, int per_group,
                                                       unsigned int *__restrict__ gsum) {
  const int r = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (r >= NR) return;
  const long t0 = (long)g * per_group, t1 = min(ntiles, t0 + per_group);
  unsigned int sacc = 0;
  for (long t = t0; t < t1; t++) sacc += hist[(size_t)t * NR + r];
  gsum[(size_t)g * NR + r] = sacc;
}

// one workgroup: per root the exclusive scan over the groups, the roots' totals and THEIR exclusive scan = root_start; gsum becomes
// the first place of the group's points of the root
__global__ __launch_bounds__(1024) void k_ms_bases(unsigned int *__restrict__ gsum, int NR, int groups, long n, unsigned int *__restrict__ root_start) {
  __shared__ unsigned int wsum[16], carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int r0 = 0; r0 < NR; r0 += 1024) {
    const int r = r0 + tid;
    unsigned int tot = 0;
    if (r < NR) for (int g = 0; g < groups; g++) { const unsigned int v = gsum[(size_t)g * NR + r]; gsum[(size_t)g * NR + r] = tot; tot += v; }
    unsigned int x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    unsigned int base = carry;
    for (int w = 0; w < wv; w++) base += wsum[w];
    const unsigned int start = base + x - tot;
    if (r < NR) {
      root_start[r] = start;
      for (int g = 0; g < groups; g++) gsum[(size_t)g * NR + r] += start;
    }
    __syncthreads();
    if (tid == 1023) carry = base + x;
    __syncthreads();
  }
  if (tid == 0) root_start[NR] = (unsigned int)n;
}

// hist[tile][r] -> the first place of tile `tile`'s points of root r
__global__ __launch_bounds__(256) void k_ms_tile_offsets(unsigned int *__restrict__ hist, int NR, long ntiles, int per_group,
                                                         const unsigned int *__restrict__ gsum) {
  const int r = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (r >= NR) return;
  const long t0 = (long)g * per_group, t1 = min(ntiles, t0 + per_group);
  unsigned int run = gsum[(size_t)g * NR + r];
  for (long t = t0; t < t1; t++) { const unsigned int c = hist[(size_t)t * NR + r]; hist[(size_t)t * NR + r] = run; run += c; }
}

// every point's record to its place in root order (stable); ck0 = (root, scan), the tags again as a dense stream, optionally the index.
// 256 points per round, the next round's loads in flight under the current round's three barriers (a barrier-free variant -- every
// wavefront walking a quarter of the tile on its own running places -- measured slower: 312 against 223 us, one serial chain per wavefront).
__global__ __launch_bounds__(256) void k_ms_scatter(const float *__restrict__ xyz, const unsigned short *__restrict__ tag,
                                                    const unsigned short *__restrict__ rid, long n, const unsigned int *__restrict__ off, int NR,
                                                    int rbits, int fb, uint4 *__restrict__ rec, unsigned int *__restrict__ ck0,
                                                    unsigned short *__restrict__ tags, unsigned int *__restrict__ idxs) {
  __shared__ unsigned int run[MS_MAX_ROOTS];
  __shared__ unsigned short wc[4][MS_MAX_ROOTS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned int *orow = off + (size_t)blockIdx.x * NR;
  for (int t = tid; t < NR; t += 256) { run[t] = orow[t]; wc[0][t] = wc[1][t] = wc[2][t] = wc[3][t] = 0; }
  const long a = (long)blockIdx.x * MS_TILE, e = min(n, a + MS_TILE);
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned int r = 0, t = 0;
  float x = 0, y = 0, z = 0;
  if (a + tid < e) { const long p = a + tid; r = rid[p]; t = tag[p]; x = xyz[3 * p]; y = xyz[3 * p + 1]; z = xyz[3 * p + 2]; }
  __syncthreads();
  for (long p0 = a; p0 < e; p0 += 256) {
    const long p = p0 + tid, pn = p + 256;
    const bool act = p < e;
    unsigned int rn = 0, tn = 0;
    float xn = 0, yn = 0, zn = 0;
    if (pn < e) { rn = rid[pn]; tn = tag[pn]; xn = xyz[3 * pn]; yn = xyz[3 * pn + 1]; zn = xyz[3 * pn + 2]; }
    unsigned long long m = __ballot(act);                    // lanes of this wavefront with the same root
    for (int b = 0; b < rbits; b++) {
      const unsigned long long bal = __ballot(act && ((r >> b) & 1u));
      m &= ((r >> b) & 1u) ? bal : ~bal;
    }
    const unsigned int below = (unsigned int)__popcll(m & lt), cnt = (unsigned int)__popcll(m);
    if (act && below == 0) wc[wv][r] = (unsigned short)cnt;
    __syncthreads();
    unsigned int d = 0;
    if (act) {
      d = run[r] + below;
      for (int w = 0; w < wv; w++) d += wc[w][r];
    }
    __syncthreads();
    if (act && below == 0) { atomicAdd(&run[r], cnt); wc[wv][r] = 0; }
    if (act) {
      rec[d] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), t);
      ck0[d] = (r << fb) | (t & 511u);
      tags[d] = (unsigned short)t;
      if (idxs) idxs[d] = (unsigned int)p;
    }
    r = rn; t = tn; x = xn; y = yn; z = zn;
    __syncthreads();
  }
}

// the sorted path's 64-bit sort values (point << 15 | octants << 9 | scan) in root order, from the fast path's root sort
__global__ void k_vals_from_tags(const unsigned int *__restrict__ idxs, const unsigned short *__restrict__ tag, long n,
                                 unsigned long long *__restrict__ vals) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long p = idxs[i];
  vals[i] = (p << 15) | (unsigned long long)tag[p];
}

__global__ void k_root_starts(const unsigned int *__restrict__ keys_sorted, const unsigned int *__restrict__ rootid_incl, long n, long NR,
                              unsigned int *__restrict__ root_start) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (i == 0 || keys_sorted[i] != keys_sorted[i - 1])) root_start[rootid_incl[i] - 1] = (unsigned int)i;
  if (i == 0) root_start[NR] = (unsigned int)n;
}

__global__ void k_root_tiles(const unsigned int *__restrict__ root_start, long NR, unsigned int *__restrict__ tiles) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < NR) tiles[r] = (root_start[r + 1] - root_start[r] + PART_TS - 1) / PART_TS;
  else if (r == NR) tiles[r] = 0;
}

// recut (bavoxel.hpp:737-776) subdivides a root voxel only when it holds more than min_ps points and is NOT a plane: levels 1 and 2
// exist for the points of those roots alone.  On the shipped window 77.5 % of the points sit in root voxels that ARE planes
// (profiles/r06_assoc_levels.txt): their records are neither partitioned nor summed again.  One workgroup walks the roots (thousands,
// not millions): tile_base[r] = partition tiles in front of root r, live_start[r] = records of split roots in front of it (the root's
// place in the compact level lists); plan = {tiles, live records}.
__global__ __launch_bounds__(1024) void k_live_plan(const unsigned int *__restrict__ root_start, const unsigned char *__restrict__ status0, long NR,
                                                    unsigned int *__restrict__ tile_base, unsigned int *__restrict__ live_start,
                                                    unsigned int *__restrict__ plan) {
  __shared__ unsigned int wsum[2][16], carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (long r0 = 0; r0 <= NR; r0 += 1024) {
    const long r = r0 + tid;
    unsigned int cnt = 0, til = 0;
    if (r < NR && status0[r] == 2 /* NODE_SPLIT */) { cnt = root_start[r + 1] - root_start[r]; til = (cnt + PART_TS - 1) / PART_TS; }
    unsigned int xc = cnt, xt = til;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int yc = __shfl_up(xc, d, 64), yt = __shfl_up(xt, d, 64);
      if (lane >= d) { xc += yc; xt += yt; }
    }
    if (lane == 63) { wsum[0][wv] = xc; wsum[1][wv] = xt; }
    __syncthreads();
    unsigned int bc = carry[0], bt = carry[1];
    for (int w = 0; w < wv; w++) { bc += wsum[0][w]; bt += wsum[1][w]; }
    if (r <= NR) { live_start[r] = bc + xc - cnt; tile_base[r] = bt + xt - til; }
    __syncthreads();
    if (tid == 1023) { carry[0] = bc + xc; carry[1] = bt + xt; }
    __syncthreads();
  }
  if (tid == 0) { plan[0] = carry[1]; plan[1] = carry[0]; }
}

// the root of tile t: the last r with tile_base[r] <= t
__device__ __forceinline__ int part_tile_root(const unsigned int *__restrict__ tile_base, int NR, unsigned int t) {
  int lo = 0, hi = NR - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_base[mid] <= t) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// per tile: how many of its records fall into each of the 64 (o1, o2) cells
__global__ __launch_bounds__(256) void k_part_hist(const unsigned short *__restrict__ tags, const unsigned int *__restrict__ root_start,
                                                   const unsigned int *__restrict__ tile_base, int NR, unsigned int *__restrict__ hist) {
  __shared__ unsigned int h[64];
  const unsigned int t = blockIdx.x;
  const int r = part_tile_root(tile_base, NR, t);
  const unsigned int a = root_start[r] + (t - tile_base[r]) * PART_TS, e = min(a + PART_TS, root_start[r + 1]);
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  for (unsigned int j = a + threadIdx.x; j < e; j += 256) atomicAdd(&h[(tags[j] >> 9) & 63u], 1u);
  __syncthreads();
  if (threadIdx.x < 64) hist[(size_t)t * 64 + threadIdx.x] = h[threadIdx.x];
}

// per root (one wavefront, lane = cell): where each tile's records of each cell go, for the 64-way (level 2) and the 8-way (level 1)
// partition -- cell-major inside the root, tile order inside a cell (= the stable order)
__global__ __launch_bounds__(64) void k_part_offsets(const unsigned int *__restrict__ hist, const unsigned int *__restrict__ root_start /* the roots' first places in the OUTPUT lists (k_live_plan's live_start) */,
                                                     const unsigned int *__restrict__ tile_base, int NR, unsigned int *__restrict__ off1,
                                                     unsigned int *__restrict__ off2) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r >= NR) return;
  const unsigned int t0 = tile_base[r], t1 = tile_base[r + 1];
  unsigned int tot2 = 0, tot1 = 0;
  for (unsigned int t = t0; t < t1; t++) {
    const unsigned int c = hist[(size_t)t * 64 + lane];
    unsigned int g = c;                              // sum over the lane's group of eight cells (one level-1 child) -> lanes 8k
    g += __shfl_down(g, 1, 64); g += __shfl_down(g, 2, 64); g += __shfl_down(g, 4, 64);
    tot2 += c; tot1 += g;
  }
  // exclusive prefix over the cells (lanes), and over the children (lanes 0, 8, 16, ...)
  unsigned int inc2 = tot2;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned int v = __shfl_up(inc2, d, 64); if (lane >= d) inc2 += v; }
  unsigned int run2 = root_start[r] + inc2 - tot2;
  unsigned int g1 = (lane & 7) == 0 ? tot1 : 0u, inc1 = g1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned int v = __shfl_up(inc1, d, 64); if (lane >= d) inc1 += v; }
  unsigned int run1 = root_start[r] + inc1 - g1;     // (meaningful on lanes 8k)
  for (unsigned int t = t0; t < t1; t++) {
    const unsigned int c = hist[(size_t)t * 64 + lane];
    unsigned int g = c;
    g += __shfl_down(g, 1, 64); g += __shfl_down(g, 2, 64); g += __shfl_down(g, 4, 64);
    off2[(size_t)t * 64 + lane] = run2; run2 += c;
    if ((lane & 7) == 0) { off1[(size_t)t * 8 + (lane >> 3)] = run1; run1 += g; }
  }
}

// per tile: every record to its place in the level-1 and the level-2 list, with that level's composite key (root, octants, scan)
// beside it.  Stable: a record's rank among the tile's records of its cell = cell counts of the earlier rounds + of the earlier
// wavefronts of this round + the lower lanes of its wavefront with the same cell (ballots over the cell's bits).
__global__ __launch_bounds__(256) void k_part_scatter(const uint4 *__restrict__ rec, const unsigned int *__restrict__ idx0,
                                                      const unsigned int *__restrict__ root_start, const unsigned int *__restrict__ tile_base,
                                                      int NR, const unsigned int *__restrict__ off1, const unsigned int *__restrict__ off2,
                                                      int fb, int levels, uint4 *__restrict__ rec1, unsigned int *__restrict__ ck1,
                                                      unsigned int *__restrict__ idx1, uint4 *__restrict__ rec2,
                                                      unsigned int *__restrict__ ck2, unsigned int *__restrict__ idx2) {
  __shared__ unsigned int run2[64], run1[8], wc2[4][64], wc1[4][8];
  const unsigned int t = blockIdx.x;
  const int r = part_tile_root(tile_base, NR, t);
  const unsigned int a = root_start[r] + (t - tile_base[r]) * PART_TS, e = min(a + PART_TS, root_start[r + 1]);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 64) { run2[tid] = off2[(size_t)t * 64 + tid]; wc2[0][tid] = wc2[1][tid] = wc2[2][tid] = wc2[3][tid] = 0; }
  if (tid < 8) { run1[tid] = off1[(size_t)t * 8 + tid]; wc1[0][tid] = wc1[1][tid] = wc1[2][tid] = wc1[3][tid] = 0; }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (unsigned int j0 = a; j0 < e; j0 += 256) {
    const unsigned int j = j0 + tid;
    const bool act = j < e;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (act) q = rec[j];
    const unsigned int cell = (q.w >> 9) & 63u, child = cell >> 3;
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int b = 5; b >= 3; b--) {
      const unsigned long long bal = __ballot(act && ((cell >> b) & 1u));
      m &= ((cell >> b) & 1u) ? bal : ~bal;
    }
    const unsigned long long m1 = m;
#pragma unroll
    for (int b = 2; b >= 0; b--) {
      const unsigned long long bal = __ballot(act && ((cell >> b) & 1u));
      m &= ((cell >> b) & 1u) ? bal : ~bal;
    }
    const unsigned int r1 = __popcll(m1 & lt), r2 = __popcll(m & lt);
    if (act && r1 == 0) wc1[wv][child] = __popcll(m1);
    if (act && r2 == 0) wc2[wv][cell] = __popcll(m);
    __syncthreads();
    unsigned int d1 = 0, d2 = 0;
    if (act) {
      d1 = run1[child] + r1; d2 = run2[cell] + r2;
      for (int w = 0; w < wv; w++) { d1 += wc1[w][child]; d2 += wc2[w][cell]; }
    }
    __syncthreads();
    if (tid < 64) { run2[tid] += wc2[0][tid] + wc2[1][tid] + wc2[2][tid] + wc2[3][tid]; wc2[0][tid] = wc2[1][tid] = wc2[2][tid] = wc2[3][tid] = 0; }
    if (tid < 8) { run1[tid] += wc1[0][tid] + wc1[1][tid] + wc1[2][tid] + wc1[3][tid]; wc1[0][tid] = wc1[1][tid] = wc1[2][tid] = wc1[3][tid] = 0; }
    if (act) {
      const unsigned int fr = q.w & 511u;
      rec1[d1] = q;
      ck1[d1] = ((((unsigned int)r << 3) | child) << fb) | fr;
      if (idx1) idx1[d1] = idx0[j];
      if (levels > 2) {
        rec2[d2] = q;
        ck2[d2] = ((((unsigned int)r << 6) | cell) << fb) | fr;
        if (idx2) idx2[d2] = idx0[j];
      }
    }
    __syncthreads();
  }
}

// k_seg_clusters on the records themselves (no index, no gather): same sums in the same order, as two launches.
//   k_seg_lane  one lane per segment: segments of up to SEG_LANE_MAX points are summed by their lane (a loop over its contiguous
//               records, the next one prefetched); longer ones are appended to a work list (the very long ones to a second list that
//               is served first).  With one WAVEFRONT per segment, as k_seg_clusters has it, level 2 of the shipped window launched
//               792 000 wavefronts of which nine in ten found a short segment and left: 290 us of launch throughput, not of work.
//   k_seg_wave  persistent wavefronts take the listed segments in turn: 64 points at a time are expanded into their 18 terms in LDS
//               (the chunk padded with +0.0: x + 0.0 = x bit for bit for every x these sums can hold), and lanes 0..17 each add one term
//               column in order -- a fixed, fully unrolled chain of 64 additions whose LDS reads run a batch ahead of the adds
//               (the rolled loop with its tail paid a full LDS round trip per 8 additions: ~1.5 us per chunk).
constexpr int SEG_LANE_MAX = 32;
constexpr unsigned int SEG_VERY_LONG = 2048;

__global__ __launch_bounds__(256) void k_seg_lane(const uint4 *__restrict__ rec, const double *__restrict__ poses,
                                                  const unsigned int *__restrict__ seg_start, const unsigned long long *__restrict__ seg_ck,
                                                  long NS, double *__restrict__ seg_body, double *__restrict__ seg_world,
                                                  unsigned int *__restrict__ counters /* [0] long, [1] very long */,
                                                  uint4 *__restrict__ long_list, uint4 *__restrict__ vlong_list /* {segment, first, end, scan} */) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned int i0 = 0, i1 = 0;
  if (s < NS) { i0 = seg_start[s]; i1 = seg_start[s + 1]; }
  const unsigned int cnt = i1 - i0;
  // the long ones join their list: one atomic per wavefront and class
  const bool is_long = cnt > (unsigned int)SEG_LANE_MAX, is_vlong = cnt > SEG_VERY_LONG;
  {
    const unsigned long long ml = __ballot(is_long && !is_vlong), mv = __ballot(is_vlong);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned int bl = 0, bv = 0;
    if (lane == 0) {
      if (ml) bl = atomicAdd(&counters[0], (unsigned int)__popcll(ml));
      if (mv) bv = atomicAdd(&counters[1], (unsigned int)__popcll(mv));
    }
    bl = __shfl(bl, 0, 64); bv = __shfl(bv, 0, 64);
    if (is_long) {
      const uint4 d = make_uint4((unsigned int)s, i0, i1, (unsigned int)(seg_ck[s] & 511ull));
      if (is_vlong) vlong_list[bv + __popcll(mv & lt)] = d; else long_list[bl + __popcll(ml & lt)] = d;
    }
  }
  if (s >= NS || is_long) return;
  const double *pose = poses + 12 * (long)(seg_ck[s] & 511ull);
  double P[12];
#pragma unroll
  for (int c = 0; c < 12; c++) P[c] = pose[c];
  double acc[SEG_TERMS];
#pragma unroll
  for (int c = 0; c < SEG_TERMS; c++) acc[c] = 0.0;
  uint4 qn = make_uint4(0, 0, 0, 0);
  if (cnt) qn = rec[i0];
  for (unsigned int u = 0; u < cnt; u++) {
    const float x[3] = {__uint_as_float(qn.x), __uint_as_float(qn.y), __uint_as_float(qn.z)};
    if (u + 1 < cnt) qn = rec[i0 + u + 1];
    const PointTerms t = point_terms(x, P);
#pragma unroll
    for (int c = 0; c < SEG_TERMS; c++) acc[c] = __dadd_rn(acc[c], t.t[c]);
  }
#pragma unroll
  for (int c = 0; c < 9; c++) { seg_body[s * 10 + c] = acc[c]; seg_world[s * 10 + c] = acc[9 + c]; }
  seg_body[s * 10 + 9] = seg_world[s * 10 + 9] = (double)cnt;
}

// One listed segment per wavefront at a time, four wavefronts per workgroup.  The split is static but not blind: the first
// min(nv, C) wavefronts take the very long segments (> 2 048 points: each is most of a wavefront's fair share already), with issue
// priority, and NOTHING else; the long ones are dealt round among the other wavefronts.  (A blind round-robin left the wavefront that
// drew the 8 255-point segment with its full share of the others as well: 175 chunks against a mean of 50, 234 us per level.  A
// shared cursor balances perfectly and costs more than it saves: 100 000 atomics on one address serialise in the L2 -- 0.7-1.1 ms
// per level, whether the answer is prefetched two segments ahead or not.)
// Nothing in the loop waits for a dependent global load: an entry is a 16-byte descriptor, the next one is loaded while the current
// segment runs, the next segment's first records during the current one's last chunk, and the pose comes through the scalar cache
// (its scan index is made wave-uniform).  The add phase runs with lanes 0..17 only.
// (Also measured, round 5: three segments per wavefront, lanes 18 k .. 18 k + 17 adding slot k's chunk -- one add phase for three chunks,
// but one wavefront per SIMD and a very long segment that shares its wavefront with two others: 522 / 404 / 192 us per level against
// 216 / 209 / 138.)
__global__ __launch_bounds__(256) void k_seg_wave(const uint4 *__restrict__ rec, const double *__restrict__ poses,
                                                  double *__restrict__ seg_body, double *__restrict__ seg_world,
                                                  const unsigned int *__restrict__ counters /* [0] long, [1] very long */,
                                                  const uint4 *__restrict__ long_list, const uint4 *__restrict__ vlong_list) {
  __shared__ double lds[4][SEG_TERMS * SEG_LD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned int nl = counters[0], nv = counters[1];
  // consumer number: wavefront (b, w) is number b + B ((w - b) mod 4) -- the first B numbers are ONE wavefront of every workgroup, on a
  // SIMD that rotates with b.  With the plain b * 4 + w the 914 very long segments of the shipped window's level 0 went to the four
  // wavefronts of the first 229 workgroups: four 100-chunk chains on one SIMD, each advancing at a quarter of its speed (the PMC
  // counters: the mean wavefront is resident 60 % of the launch, the VALU 12 % busy per wavefront -- profiles/r05q_association_pmc.csv)
  const unsigned int NC = gridDim.x * 4, me = blockIdx.x + gridDim.x * ((wv - blockIdx.x) & 3u);
  // this wavefront's list, first entry and stride
  const unsigned int heavy = nv < NC ? nv : NC;              // wavefronts that carry very long segments
  const bool mine_heavy = me < heavy;
  const bool all_heavy = heavy == NC;                          // (more very long segments than wavefronts: everybody takes both kinds)
  const uint4 *list = mine_heavy ? vlong_list : long_list;
  unsigned int total = mine_heavy ? nv : nl;
  unsigned int ent = mine_heavy ? me : me - heavy;
  unsigned int stride = mine_heavy ? NC : NC - heavy;
  bool second_pass = false;                                    // all_heavy: after the very long ones, the long ones dealt round everybody
  if (!mine_heavy && ent >= total) return;
  const uint4 none = make_uint4(0, 0, 0, 0);
  uint4 cur = ent < total ? list[ent] : none, nxt = ent + stride < total ? list[ent + stride] : none;
  if (mine_heavy) __builtin_amdgcn_s_setprio(3);
  uint4 qn = rec[min(cur.y + lane, cur.z - 1)];
  double *mine = lds[wv];
  const double *colp = mine + min(lane, SEG_TERMS - 1) * SEG_LD;
  for (;;) {
    if (cur.z <= cur.y) {                                      // this list is through
      if (!(all_heavy && !second_pass)) break;
      second_pass = true;
      __builtin_amdgcn_s_setprio(0);
      list = long_list; total = nl; ent = me; stride = NC;
      cur = ent < total ? list[ent] : none; nxt = ent + stride < total ? list[ent + stride] : none;
      if (cur.z <= cur.y) break;
      qn = rec[min(cur.y + lane, cur.z - 1)];
    }
    const double *pose = poses + 12 * (size_t)__builtin_amdgcn_readfirstlane((int)cur.w);
    double P[12];
#pragma unroll
    for (int c = 0; c < 12; c++) P[c] = pose[c];
    double acc = 0.0;
    for (unsigned int i = cur.y; i < cur.z; i += 64) {
      const bool in = i + lane < cur.z;
      const float x[3] = {__uint_as_float(qn.x), __uint_as_float(qn.y), __uint_as_float(qn.z)};
      if (i + 64 < cur.z) qn = rec[min(i + 64 + lane, cur.z - 1)];
      else if (nxt.z > nxt.y) qn = rec[min(nxt.y + lane, nxt.z - 1)];      // the next segment's first chunk
      const PointTerms t = point_terms(x, P);
      if (i + 64 <= cur.z) {                     // (a full chunk, the common case: no padding selects -- 36 of the loop's ~190 instructions)
#pragma unroll
        for (int c = 0; c < SEG_TERMS; c++) mine[c * SEG_LD + lane] = t.t[c];
      } else {
#pragma unroll
        for (int c = 0; c < SEG_TERMS; c++) mine[c * SEG_LD + lane] = in ? t.t[c] : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the wave's LDS writes have landed
      if (lane < SEG_TERMS) {
        double va[16], vb[16];
#pragma unroll
        for (int u = 0; u < 16; u++) va[u] = colp[u];
#pragma unroll
        for (int b = 0; b < 4; b++) {
          if (b < 3) {
#pragma unroll
            for (int u = 0; u < 16; u++) vb[u] = colp[16 * (b + 1) + u];
          }
#pragma unroll
          for (int u = 0; u < 16; u++) acc = __dadd_rn(acc, va[u]);
#pragma unroll
          for (int u = 0; u < 16; u++) va[u] = vb[u];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const size_t s = cur.x;
    if (lane < 9) seg_body[s * 10 + lane] = acc;
    else if (lane < SEG_TERMS) seg_world[s * 10 + lane - 9] = acc;
    else if (lane == SEG_TERMS) seg_body[s * 10 + 9] = seg_world[s * 10 + 9] = (double)(cur.z - cur.y);
    ent += stride;
    cur = nxt;
    nxt = ent + stride < total ? list[ent + stride] : none;
  }
}

inline void launch_seg_clusters(hipStream_t st, const float *xyz, const double *poses, const unsigned int *idx, const unsigned int *seg_start,
                                const unsigned long long *seg_ck, long NS_or_bound, double *seg_body, double *seg_world,
                                const unsigned int *ns_dev = nullptr) {
  if (NS_or_bound <= 0) return;
  const int sb = (int)((NS_or_bound + 255) / 256), lb = (int)((NS_or_bound + 3) / 4);
  hipLaunchKernelGGL(k_seg_clusters, dim3(sb + lb), dim3(256), 0, st, xyz, poses, idx, seg_start, seg_ck, NS_or_bound, sb, seg_body, seg_world, ns_dev);
}

// lists: [4] counters (long, very long), then two descriptor lists of NS entries each (scratch of the caller: 16 + 32 NS bytes)
inline void launch_seg_clusters_rec(hipStream_t st, const uint4 *rec, const double *poses, const unsigned int *seg_start,
                                    const unsigned long long *seg_ck, long NS, double *seg_body, double *seg_world, unsigned int *lists) {
  if (NS <= 0) return;
  unsigned int *counters = lists;
  uint4 *long_list = reinterpret_cast<uint4 *>(lists + 4), *vlong_list = long_list + NS;
  // (counters: zeroed by k_seg_heads, the launch in front of this one)
  hipLaunchKernelGGL(k_seg_lane, dim3((unsigned int)((NS + 255) / 256)), dim3(256), 0, st, rec, poses, seg_start, seg_ck, NS, seg_body, seg_world,
                     counters, long_list, vlong_list);
  const long want = (NS + 3) / 4;
  hipLaunchKernelGGL(k_seg_wave, dim3((unsigned int)(want < 1024 ? want : 1024)), dim3(256), 0, st, rec, poses, seg_body, seg_world,
                     counters, long_list, vlong_list);
}

// tree links over a level's sorted segment list: node -> first segment, node -> parent node,
// parent node -> first child node
__global__ void k_level_heads(const unsigned long long *__restrict__ seg_ck, const unsigned int *__restrict__ nid_incl,
                              const unsigned int *__restrict__ pid_incl, long NS, int parent_shift,
                              unsigned int *__restrict__ node_seg, unsigned int *__restrict__ node_parent) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= NS) return;
  if (s == NS - 1) node_seg[nid_incl[s]] = (unsigned int)NS;        // the entry behind the last node's
  const unsigned long long ck = seg_ck[s];
  if (s == 0 || (ck >> 9) != (seg_ck[s - 1] >> 9)) {
    const unsigned int j = nid_incl[s] - 1;
    node_seg[j] = (unsigned int)s;
    node_parent[j] = parent_shift ? pid_incl[s] - 1 : (unsigned int)(ck >> 15);
  }
}

__global__ void k_set_u32(unsigned int *p, unsigned int v) { *p = v; }

struct NodeTot {
  double c[10];        // world cluster over all scans (judge_eigen)
  double fixc[10];     // world cluster of the marginalised scans (fix_point after to_margi)
  double nrest;        // points in the scans that stay
  int nobs;            // scans that stay and observe the node
};

// (node totals: the per-frame world clusters added in frame order = judge_eigen's `covMat += sig_tran[i]`; 16 lanes per node, lane c < 10
// owns component c -- k_node_totals_status below)
// eigenvalues of a symmetric 3x3 (cyclic Jacobi, same rotation order as the host restatement the tests compare it with)
__device__ void eigvals3(double a00, double a01, double a02, double a11, double a12, double a22, double lam[3]) {
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12, dia = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
#define BALM_ROT(app, aqq, apq, arp, arq)                                                                   \
  if (apq != 0.0) {                                                                                         \
    const double theta = (aqq - app) / (2.0 * apq);                                                         \
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));                 \
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                                    \
    app -= t * apq; aqq += t * apq; apq = 0.0;                                                              \
    const double rp = c * arp - s * arq, rq = s * arp + c * arq;                                            \
    arp = rp; arq = rq;                                                                                     \
  }
    BALM_ROT(a00, a11, a01, a02, a12)
    BALM_ROT(a00, a22, a02, a01, a12)
    BALM_ROT(a11, a22, a12, a01, a02)
#undef BALM_ROT
  }
  double x = a00, y = a11, z = a22, t;
  if (x > y) { t = x; x = y; y = t; }
  if (y > z) { t = y; y = z; z = t; }
  if (x > y) { t = x; x = y; y = t; }
  lam[0] = x; lam[1] = y; lam[2] = z;
}

// per node: 0 = too few points (recut returns, bavoxel.hpp:744), 1 = plane (judge_eigen :654-699 accepts),
// 2 = not a plane (the node is split, or dies at the last layer)
enum { NODE_DEAD = 0, NODE_PLANE = 1, NODE_SPLIT = 2 };

// eigenvectors too (rotations accumulated in the same sweeps): the strict test needs the plane normal
__device__ void eig3_vec(double a00, double a01, double a02, double a11, double a12, double a22, double lam[3], double nrm[3]) {
  double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int p = t == 2 ? 1 : 0, q = t == 0 ? 1 : 2, r = 3 - p - q;
      if (A[p][q] == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
      A[p][p] -= tt * A[p][q]; A[q][q] += tt * A[p][q]; A[p][q] = A[q][p] = 0.0;
      const double rp = c * A[r][p] - sn * A[r][q], rq = sn * A[r][p] + c * A[r][q];
      A[r][p] = A[p][r] = rp; A[r][q] = A[q][r] = rq;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double vp = c * V[k][p] - sn * V[k][q], vq = sn * V[k][p] + c * V[k][q];
        V[k][p] = vp; V[k][q] = vq;
      }
    }
  }
  int i0 = 0;
  if (A[1][1] < A[i0][i0]) i0 = 1;
  if (A[2][2] < A[i0][i0]) i0 = 2;
  double x = A[0][0], y = A[1][1], z = A[2][2], t;
  if (x > y) { t = x; x = y; y = t; }
  if (y > z) { t = y; y = z; z = t; }
  if (x > y) { t = x; x = y; y = t; }
  lam[0] = x; lam[1] = y; lam[2] = z;
  nrm[0] = V[0][i0]; nrm[1] = V[1][i0]; nrm[2] = V[2][i0];
}

__device__ __forceinline__ int node_status_of(const double c[10], float thr, const VoxParams &pr, double *__restrict__ pl);

// k_node_totals and k_node_status in ONE launch (round 6: one launch less per level): the sixteen lanes of a node form its totals as
// k_node_totals does, hand the ten components to the node's first lane (shuffles inside the 16-lane group), which judges the plane
__global__ __launch_bounds__(256) void k_node_totals_status(const double *__restrict__ seg_world, const unsigned long long *__restrict__ seg_ck,
                                                            const unsigned int *__restrict__ node_seg, long NN, int fix_frames,
                                                            NodeTot *__restrict__ tot, float thr, VoxParams pr,
                                                            unsigned char *__restrict__ status, double *__restrict__ plane) {
  const long j = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int c = threadIdx.x & 15;
  const bool live = j < NN;
  double t = 0.0;
  if (live) {
    const unsigned int s0 = node_seg[j], s1 = node_seg[j + 1];
    if (fix_frames == 0) {
      if (c < 10) {
        unsigned int sgi = s0;
        for (; sgi + 8 <= s1; sgi += 8) {             // loads of eight scans in flight, the sum stays in scan order
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = seg_world[(size_t)(sgi + u) * 10 + c];
#pragma unroll
          for (int u = 0; u < 8; u++) t = __dadd_rn(t, v[u]);
        }
        for (; sgi < s1; sgi++) t = __dadd_rn(t, seg_world[(size_t)sgi * 10 + c]);
        tot[j].c[c] = t;
        tot[j].fixc[c] = 0.0;
        if (c == 9) tot[j].nrest = t;
      } else if (c == 10) {
        tot[j].nobs = (int)(s1 - s0);
      }
    } else if (c < 10) {
      double fx = 0.0;
      for (unsigned int sgi = s0; sgi < s1; sgi++) {
        const double v = seg_world[(size_t)sgi * 10 + c];
        t = __dadd_rn(t, v);
        if ((int)(seg_ck[sgi] & 511ull) < fix_frames) fx = __dadd_rn(fx, v);
      }
      tot[j].c[c] = t;
      tot[j].fixc[c] = fx;
    } else if (c == 10) {
      double nrest = 0.0;
      int nobs = 0;
      for (unsigned int sgi = s0; sgi < s1; sgi++)
        if ((int)(seg_ck[sgi] & 511ull) >= fix_frames) { nrest += seg_world[(size_t)sgi * 10 + 9]; nobs++; }
      tot[j].nrest = nrest;
      tot[j].nobs = nobs;
    }
  }
  double cc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) cc[k] = __shfl(t, k, 16);           // (every lane of the wavefront takes part; lanes 10..15 hold junk nobody asks for)
  if (live && c == 0) status[j] = (unsigned char)node_status_of(cc, thr, pr, plane ? plane + (size_t)j * 6 : nullptr);
}

// a node's verdict from its world cluster c[10]; pl (strict mode only): normal(3), centre(3) of the candidate plane, for the distance pass
__device__ __forceinline__ int node_status_of(const double c[10], float thr, const VoxParams &pr, double *__restrict__ pl) {
  int st = NODE_DEAD;
  if ((int)c[9] > pr.min_ps) {
    const double n = c[9], cx = c[6] / n, cy = c[7] / n, cz = c[8] / n;
    double lam[3];
    const double a00 = c[0] / n - cx * cx, a01 = c[1] / n - cx * cy, a02 = c[2] / n - cx * cz, a11 = c[3] / n - cy * cy,
                 a12 = c[4] / n - cy * cz, a22 = c[5] / n - cz * cz;
    if (!pl) {
      eigvals3(a00, a01, a02, a11, a12, a22, lam);
      st = lam[0] / lam[1] < (double)thr ? NODE_PLANE : NODE_SPLIT;
    } else {
      double nrm[3];
      eig3_vec(a00, a01, a02, a11, a12, a22, lam, nrm);
      const bool ok = lam[0] / lam[1] < (double)thr && (pr.ratio21_max <= 0 || lam[2] / lam[1] < pr.ratio21_max) &&
                      (pr.lam0_max <= 0 || lam[0] < pr.lam0_max);
      st = ok ? NODE_PLANE : NODE_SPLIT;
      pl[0] = nrm[0]; pl[1] = nrm[1]; pl[2] = nrm[2]; pl[3] = cx; pl[4] = cy; pl[5] = cz;
    }
  }
  return st;
}

// strict mode: a candidate plane with a point farther than max_dis from it is not a plane (BAs_left.hpp:658-674).
// One lane per point of the level's sorted list; a node is demoted by whichever of its points finds the violation.
__global__ __launch_bounds__(256) void k_point_plane_dist(const float *__restrict__ xyz, const ScanOf scan,
                                                          const double *__restrict__ poses, const unsigned int *__restrict__ idx,
                                                          const unsigned int *__restrict__ segid_incl,
                                                          const unsigned int *__restrict__ nid_incl, long n,
                                                          const double *__restrict__ plane, double max_dis,
                                                          unsigned char *__restrict__ status) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int j = nid_incl[segid_incl[i] - 1] - 1;
  if (status[j] != NODE_PLANE) return;
  const long p = idx[i];
  double q[3], po[3];
  world_point(xyz, poses + 12 * (long)scan.find(p, scan.first), p, q, po);
  const double *pl = plane + (size_t)j * 6;
  const double d = fabs(pl[0] * (q[0] - pl[3]) + pl[1] * (q[1] - pl[4]) + pl[2] * (q[2] - pl[5]));
  if (!(d < max_dis)) status[j] = NODE_SPLIT;       // benign race: every writer stores the same value
}

// node id of every point at this level (for the point -> feature map)
__global__ void k_point_nodes(const unsigned int *__restrict__ idx, const unsigned int *__restrict__ segid_incl,
                              const unsigned int *__restrict__ nid_incl, long n, unsigned int *__restrict__ node_of_point) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) node_of_point[idx[i]] = nid_incl[segid_incl[i] - 1] - 1;
}

// recut's descent (bavoxel.hpp:737-776) + tras_opt / push_voxel (:908-929, :30-37): a node is a feature when
// every ancestor was split, it is a plane with more than min_ps points, the scans that stay hold at least min_ps
// of them and at least min_observers of those scans observe it
// Round 6: ONE launch for the (up to) three levels -- their node lists laid end to end, each followed by a zero entry -- and ONE exclusive
// scan behind it: a node's scanned value is its feature's index in the table (level 0's features first, then level 1's, then level 2's),
// the values at the three zero entries are the running feature counts.  (Three flag launches + three library scans of two launches each before.)
struct FlagLevels { long NN[3], off[3]; const NodeTot *tot[3]; };
__global__ void k_feature_flags(FlagLevels fl, int levels, const unsigned char *__restrict__ st0,
                                const unsigned char *__restrict__ st1, const unsigned char *__restrict__ st2,
                                const unsigned int *__restrict__ parent1, const unsigned int *__restrict__ parent2,
                                VoxParams pr, unsigned int *__restrict__ flag_all) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int level = 0;
  if (levels > 1 && g >= fl.off[1]) level = 1;
  if (levels > 2 && g >= fl.off[2]) level = 2;
  const long j = g - fl.off[level], NN = fl.NN[level];
  if (j > NN) return;
  unsigned int *flag = flag_all + fl.off[level];
  if (j == NN) { flag[NN] = 0u; return; }
  const NodeTot *tot = fl.tot[level];
  bool live;
  if (level == 0) live = st0[j] == NODE_PLANE;
  else if (level == 1) live = st1[j] == NODE_PLANE && st0[parent1[j]] == NODE_SPLIT;
  else {
    const unsigned int p1 = parent2[j];
    live = st2[j] == NODE_PLANE && st1[p1] == NODE_SPLIT && st0[parent1[p1]] == NODE_SPLIT;
  }
  const NodeTot &t = tot[j];
  flag[j] = (live && (int)t.nrest >= pr.min_ps && t.nobs >= pr.min_observers && t.nobs >= 1) ? 1u : 0u;
}

__global__ void k_point_features(long n, int levels, const unsigned int *__restrict__ node0, const unsigned int *__restrict__ node1,
                                 const unsigned int *__restrict__ node2, const unsigned int *__restrict__ flag0,
                                 const unsigned int *__restrict__ flag1, const unsigned int *__restrict__ flag2,
                                 const unsigned int *__restrict__ fid0, const unsigned int *__restrict__ fid1,
                                 const unsigned int *__restrict__ fid2, unsigned int base1, unsigned int base2,
                                 int *__restrict__ feat_of_point) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int f = -1;
  unsigned int j = node0[p];
  if (flag0[j]) f = (int)fid0[j];
  // (0xffffffff: the point's root voxel was not split -- it has no node at that level)
  if (f < 0 && levels > 1 && flag1) { j = node1[p]; if (j != 0xffffffffu && flag1[j]) f = (int)(base1 + fid1[j]); }
  if (f < 0 && levels > 2 && flag2) { j = node2[p]; if (j != 0xffffffffu && flag2[j]) f = (int)(base2 + fid2[j]); }
  feat_of_point[p] = f;
}

// per-(feature, pose) body clusters = the feature node's own segments (sig_orig) of the scans that stay, shifted by
// the marginalised ones: one lane per (segment, component)
// (round 6: the level's nodes ride in the same launch -- blocks [seg_blocks, ...) are k_emit_nodes' -- and fid_excl is the feature's index
// in the whole table already, k_feature_flags' one scan over all levels)
__device__ __forceinline__ void emit_node(long j, int c, const unsigned int *__restrict__ flag, const unsigned int *__restrict__ fid_excl,
                                          const NodeTot *__restrict__ tot, int layer, double *__restrict__ coe, double *__restrict__ fixout,
                                          int *__restrict__ layer_out);
__global__ __launch_bounds__(256) void k_emit_level(long NS, int seg_blocks, const unsigned int *__restrict__ nid_incl,
                                                    const unsigned int *__restrict__ flag, const unsigned int *__restrict__ fid_excl,
                                                    const unsigned long long *__restrict__ seg_ck,
                                                    const double *__restrict__ seg_body, int Wout, int fix_frames,
                                                    double *__restrict__ out, long NN, const NodeTot *__restrict__ tot, int layer,
                                                    double *__restrict__ coe, double *__restrict__ fixout, int *__restrict__ layer_out) {
  if ((int)blockIdx.x >= seg_blocks) {
    const long tn = ((long)blockIdx.x - seg_blocks) * blockDim.x + threadIdx.x;
    if ((tn >> 4) < NN) emit_node(tn >> 4, (int)(tn & 15), flag, fid_excl, tot, layer, coe, fixout, layer_out);
    return;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long s = t >> 4;
  const int c = (int)(t & 15);
  if (s >= NS || c >= 10) return;
  const unsigned int j = nid_incl[s] - 1;
  if (!flag[j]) return;
  const int fr = (int)(seg_ck[s] & 511ull) - fix_frames;
  if (fr < 0) return;
  const size_t f = fid_excl[j];
  out[(f * Wout + (size_t)fr) * 10 + c] = seg_body[(size_t)s * 10 + c];
}

// per feature: weight = sum_i N_i over the scans that stay (VOX_HESS::push_voxel, bavoxel.hpp:42-44), fix cluster =
// the marginalised scans' world cluster, octree layer
__device__ __forceinline__ void emit_node(long j, int c, const unsigned int *__restrict__ flag, const unsigned int *__restrict__ fid_excl,
                                          const NodeTot *__restrict__ tot, int layer, double *__restrict__ coe, double *__restrict__ fixout,
                                          int *__restrict__ layer_out) {
  if (!flag[j]) return;
  const size_t f = fid_excl[j];
  if (c < 10) fixout[f * 10 + c] = tot[j].fixc[c];
  else if (c == 10) { coe[f] = tot[j].nrest; layer_out[f] = layer; }
}

// bump allocator over a caller-owned arena; whatever does not fit falls back to hipMalloc for this call and
// raises `need`, so that the caller can grow the arena for the next call
struct Scratch {
  char *base; size_t cap, off = 0, need = 0;
  std::vector<void *> extra;
  bool ok = true;
  AssocMail *persist = nullptr;          // the context's persistent small state (k_scan_heads' tile words), where the caller has one
  Scratch(void *b, size_t c) : base((char *)b), cap(c) {}
  template <class T> T *get(size_t n) {
    const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
    need += bytes;
    if (off + bytes <= cap) { T *p = (T *)(base + off); off += bytes; return p; }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return nullptr; }
    extra.push_back(p);
    return (T *)p;
  }
  ~Scratch() { for (void *p : extra) hipFree(p); }
};

unsigned int last_u32_sync(hipStream_t s, const unsigned int *d, long n);
// a count the host needs before it can size the next step, through a pinned mailbox the host polls (a copy command + a stream
// synchronise cost 15-25 us on this stack, the mailbox ~5; the association reads eleven such counts)
__global__ void k_mail_u32(const unsigned int *__restrict__ src, volatile unsigned int *__restrict__ mail, unsigned int seq) {
  mail[0] = *src;
  __threadfence_system();
  mail[1] = seq;
  __threadfence_system();
}

// up to four counts in one round trip (mail[4..7]); src pointers may repeat
__global__ void k_mail_u32x4(const unsigned int *__restrict__ a, const unsigned int *__restrict__ b, const unsigned int *__restrict__ c,
                             const unsigned int *__restrict__ d, volatile unsigned int *__restrict__ mail, unsigned int seq) {
  mail[4] = *a; mail[5] = *b; mail[6] = *c; mail[7] = *d;
  __threadfence_system();
  mail[1] = seq;
  __threadfence_system();
}
bool mail_u32x4(hipStream_t s, AssocMail *mail, const unsigned int *const src[4], unsigned int out[4]) {
  if (mail && mail->host && mail->dev) {
    const unsigned int seq = ++mail->seq;
    hipLaunchKernelGGL(k_mail_u32x4, dim3(1), dim3(1), 0, s, src[0], src[1], src[2], src[3], mail->dev, seq);
    volatile unsigned int *h = mail->host;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; spin++) {
      if (h[1] == seq) { std::atomic_thread_fence(std::memory_order_acquire); for (int k = 0; k < 4; k++) out[k] = h[4 + k]; return true; }
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (hipStreamSynchronize(s) == hipSuccess && h[1] == seq) { for (int k = 0; k < 4; k++) out[k] = h[4 + k]; return true; }
    return false;
  }
  for (int k = 0; k < 4; k++) if (hipMemcpyAsync(&out[k], src[k], sizeof(unsigned int), hipMemcpyDeviceToHost, s) != hipSuccess) return false;
  return hipStreamSynchronize(s) == hipSuccess;
}

unsigned int last_u32(hipStream_t s, const unsigned int *d, long n, AssocMail *mail = nullptr) {
  if (n > 0 && mail && mail->host && mail->dev) {
    const unsigned int seq = ++mail->seq;
    hipLaunchKernelGGL(k_mail_u32, dim3(1), dim3(1), 0, s, d + n - 1, mail->dev, seq);
    volatile unsigned int *h = mail->host;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; spin++) {
      if (h[1] == seq) { std::atomic_thread_fence(std::memory_order_acquire); return h[0]; }
      if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (hipStreamSynchronize(s) == hipSuccess && h[1] == seq) return h[0];
    return 0;
  }
  return last_u32_sync(s, d, n);
}

unsigned int last_u32_sync(hipStream_t s, const unsigned int *d, long n) {
  unsigned int v = 0;
  if (n > 0) { hipMemcpyAsync(&v, d + n - 1, sizeof(v), hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
  return v;
}

void scan_incl(Scratch &sc, hipStream_t s, const unsigned int *in, unsigned int *out, long n) {
  size_t tmp = 0;
  rocprim::inclusive_scan(nullptr, tmp, in, out, (size_t)n, rocprim::plus<unsigned int>(), s);
  void *d = sc.get<char>(tmp);
  if (d) rocprim::inclusive_scan(d, tmp, in, out, (size_t)n, rocprim::plus<unsigned int>(), s);
}

// inclusive scan of the HEAD FLAGS of a sorted key list (position i starts a run: i == 0 or key[i] >> shift differs from its
// predecessor's) without the flags ever being stored: the scan reads the keys through a transform iterator -- one launch and 4 + 4
// bytes per element instead of a flag kernel (4 + 4) in front of the scan (4 + 4)
template <class K>
struct HeadFlagOf {
  const K *key; int shift;
  __device__ unsigned int operator()(long i) const { return (i == 0 || (key[i] >> shift) != (key[i - 1] >> shift)) ? 1u : 0u; }
};
// The same scan as ONE pass of our own for the lists that have an entry per POINT (the root keys and the three levels' composite keys of
// balm_associate: 13.4 M entries on the shipped window, where the library's scan -- default tuning, no gfx950 entry in its tables --
// takes 75 us = 1.4 TB/s for 4 + 4 bytes per entry).  Decoupled look-back: workgroup b scans tile b (8 192 entries), publishes its sum
// {1, sum} as ONE 64-bit word, finds its offset by looking back over its predecessors' words (256 at a time, one per thread) up to the
// nearest one that already holds an inclusive prefix {2, prefix}, publishes its own, adds, writes.  A thread owns FOUR consecutive entries
// in each of the tile's eight parts: every load and store is a full 16 (32) bytes per lane, consecutive lanes adjacent.
//   The tile is the workgroup's INDEX, not a ticket drawn from a counter (the library's way, and this kernel's first form: 3 273 atomics
// on one address cost 21 of its 72 us).  That is safe where workgroups start in index order -- what a tile waits for then already
// runs -- which is how the dispatcher of this hardware walks a one-dimensional grid, per XCD; it is not a guarantee, so the wait is
// bounded: a workgroup that has not seen a predecessor's word after SH_SPIN_LIMIT looks (~tens of ms; a normal wait is microseconds)
// COUNTS the heads in front of its tile itself (sh_count_before: correct whatever the others do, and it terminates), publishes and goes on.
constexpr int SH_BLOCK = 256, SH_PARTS = 8, SH_TILE = 4 * SH_PARTS * SH_BLOCK;
constexpr int SH_SPIN_LIMIT = 1 << 15;
__device__ __forceinline__ void sh_load4(const unsigned int *p, unsigned int (&k)[4]) {
  const uint4 v = *reinterpret_cast<const uint4 *>(p);
  k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
}
__device__ __forceinline__ void sh_load4(const unsigned long long *p, unsigned long long (&k)[4]) {
  const ulonglong2 a = reinterpret_cast<const ulonglong2 *>(p)[0], b = reinterpret_cast<const ulonglong2 *>(p)[1];
  k[0] = a.x; k[1] = a.y; k[2] = b.x; k[3] = b.y;
}
// heads among the entries [0, end): the whole workgroup, the answer in every thread (s_red: [4])
template <class K>
__device__ unsigned int sh_count_before(const K *__restrict__ key, int shift, long end, unsigned int *s_red) {
  unsigned int c = 0;
  for (long i = threadIdx.x; i < end; i += SH_BLOCK) c += (i == 0 || (key[i] >> shift) != (key[i - 1] >> shift)) ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
template <class K>
__global__ __launch_bounds__(SH_BLOCK) void k_scan_heads(const K *__restrict__ key, int shift, unsigned int *__restrict__ out, long n,
                                                         unsigned long long *__restrict__ state /* [tiles]: {generation : 30, kind : 2, value : 32} */,
                                                         unsigned long long gen /* this scan's generation << 34 */, int spin_limit) {
  __shared__ unsigned int s_look[4], s_found[4], s_wsum[SH_PARTS][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned int tile = blockIdx.x;
  const long t0 = (long)tile * SH_TILE;
  const bool full = t0 + SH_TILE <= n;
  unsigned int v[SH_PARTS][4], tsum[SH_PARTS];         // [part][entry]: inclusive within the thread's four; the thread's sum per part
#pragma unroll
  for (int q = 0; q < SH_PARTS; q++) {
    const long i0 = t0 + (long)q * (4 * SH_BLOCK) + 4 * tid;
    K k[4];
    if (full) {
      sh_load4(key + i0, k);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) k[j] = i0 + j < n ? key[i0 + j] : (K)0;
    }
    // the entry in front of the thread's four: the left neighbour's last (lane 0: from memory)
    K kp = (K)__shfl_up(k[3], 1, 64);
    if (lane == 0) kp = (i0 > 0 && i0 - 1 < n) ? key[i0 - 1] : (K)0;
    unsigned int run = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const K prev = j == 0 ? kp : k[j - 1];
      const bool head = i0 + j < n && (i0 + j == 0 || (k[j] >> shift) != (prev >> shift));
      run += head ? 1u : 0u;
      v[q][j] = run;
    }
    tsum[q] = run;
  }
  // exclusive offsets of the thread's parts inside the tile: wave scans, wave sums through LDS
  unsigned int wex[SH_PARTS];
#pragma unroll
  for (int q = 0; q < SH_PARTS; q++) {
    unsigned int x = tsum[q];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    wex[q] = x - tsum[q];
    if (lane == 63) s_wsum[q][wv] = x;
  }
  __syncthreads();
  unsigned int off[SH_PARTS], total = 0;
#pragma unroll
  for (int q = 0; q < SH_PARTS; q++) {
    unsigned int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const unsigned int sw = s_wsum[q][w]; if (w < wv) before += sw; all += sw; }
    off[q] = total + before + wex[q];
    total += all;
  }
  // the tile's offset.  All 256 threads look back, one predecessor each.
  unsigned int prefix = 0;
  if (tile == 0) {
    if (tid == 0) __hip_atomic_store(state, gen | (2ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (tid == 0) __hip_atomic_store(state + tile, gen | (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long pos = (long)tile - 1;
    bool gave_up = false;
    for (;;) {
      const long t = pos - tid;
      unsigned long long w = 2ull << 32;                                       // (in front of tile 0: an inclusive prefix of zero)
      bool late = false;
      if (t >= 0) {                                                            // a word of another generation is an older scan's: not there yet
        int spins = 0;
        do {
          w = __hip_atomic_load(state + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w = (w >> 34) == (gen >> 34) ? w & ((1ull << 34) - 1ull) : 0ull;
        } while ((w >> 32) == 0 && ++spins < spin_limit);
        late = (w >> 32) == 0;
      }
      const unsigned long long incl = __ballot((w >> 32) == 2ull);
      const int upto = incl ? __builtin_ctzll(incl) : 63;                      // this wavefront's nearest predecessor with an inclusive prefix
      const unsigned long long lates = __ballot(late && lane <= upto);         // (a word behind that one is not needed)
      unsigned int x = lane <= upto ? (unsigned int)w : 0u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
      if (lane == 0) { s_look[wv] = x; s_found[wv] = (incl ? 1u : 0u) | (lates ? 2u : 0u); }
      __syncthreads();
      bool done = false;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!done) { prefix += s_look[k]; done = (s_found[k] & 1u) != 0u; gave_up = gave_up || (s_found[k] & 2u) != 0u; }
      }
      __syncthreads();
      if (done || gave_up) break;
      pos -= SH_BLOCK;
    }
    if (gave_up) prefix = sh_count_before(key, shift, t0, s_look);
    if (tid == 0) __hip_atomic_store(state + tile, gen | (2ull << 32) | (unsigned long long)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int q = 0; q < SH_PARTS; q++) {
    const long i0 = t0 + (long)q * (4 * SH_BLOCK) + 4 * tid;
    const unsigned int b = prefix + off[q];
    if (full) {
      *reinterpret_cast<uint4 *>(out + i0) = make_uint4(b + v[q][0], b + v[q][1], b + v[q][2], b + v[q][3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) if (i0 + j < n) out[i0 + j] = b + v[q][j];
    }
  }
}

template <class K>
void scan_heads(Scratch &sc, hipStream_t s, const K *key, int shift, unsigned int *out, long n) {
  if (n >= SH_TILE && n < (1l << 32)) {                // (the counts fit 32 bits; key and out come from the arena: 256-byte aligned)
    const long tiles = (n + SH_TILE - 1) / SH_TILE;
    if ((((uintptr_t)key | (uintptr_t)out) & 15) == 0) {
      const char *e = getenv("BALM_SCAN_SPIN");        // (tests: 1 = every workgroup gives up at its first look and counts for itself)
      const int spin_limit = e && atoi(e) > 0 ? atoi(e) : SH_SPIN_LIMIT;
      unsigned long long *state = nullptr, gen = 1;
      if (sc.persist && sc.persist->scan_state && tiles <= SCAN_TILES_CAP) {
        // the context's persistent words, told apart by the scan's generation: nothing to clear (30 bits: cleared once per 2^30 scans)
        state = sc.persist->scan_state;
        if (++sc.persist->scan_gen >= (1u << 30)) { hipMemsetAsync(state, 0, SCAN_TILES_CAP * sizeof(unsigned long long), s); sc.persist->scan_gen = 1; }
        gen = sc.persist->scan_gen;
      } else if ((state = sc.get<unsigned long long>((size_t)tiles)) != nullptr) {
        hipMemsetAsync(state, 0, (size_t)tiles * sizeof(unsigned long long), s);
      }
      if (state) {
        hipLaunchKernelGGL((k_scan_heads<K>), dim3((unsigned int)tiles), dim3(SH_BLOCK), 0, s, key, shift, out, n, state, gen << 34, spin_limit);
        return;
      }
    }
  }
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<long>(0), HeadFlagOf<K>{key, shift});
  size_t tmp = 0;
  rocprim::inclusive_scan(nullptr, tmp, in, out, (size_t)n, rocprim::plus<unsigned int>(), s);
  void *d = sc.get<char>(tmp);
  if (d) rocprim::inclusive_scan(d, tmp, in, out, (size_t)n, rocprim::plus<unsigned int>(), s);
}

void scan_excl(Scratch &sc, hipStream_t s, const unsigned int *in, unsigned int *out, long n) {
  size_t tmp = 0;
  rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, (size_t)n, rocprim::plus<unsigned int>(), s);
  void *d = sc.get<char>(tmp);
  if (d) rocprim::exclusive_scan(d, tmp, in, out, 0u, (size_t)n, rocprim::plus<unsigned int>(), s);
}

template <class K, class V>
void sort_pairs(Scratch &sc, hipStream_t s, const K *kin, K *kout, const V *vin, V *vout, long n, int end_bit, int begin_bit = 0) {
  size_t tmp = 0;
  rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, (size_t)n, begin_bit, end_bit, s);
  void *d = sc.get<char>(tmp);
  if (d) rocprim::radix_sort_pairs(d, tmp, kin, kout, vin, vout, (size_t)n, begin_bit, end_bit, s);
}

inline int grid_for(long n, int bs) { return (int)((n + bs - 1) / bs); }

struct Level {
  long NS = 0, NN = 0;
  unsigned long long *seg_ck = nullptr;
  double *seg_body = nullptr;
  unsigned int *node_seg = nullptr, *node_parent = nullptr, *flag = nullptr, *fid = nullptr, *seg_node = nullptr;
  unsigned char *status = nullptr;
  NodeTot *tot = nullptr;
};

}  // namespace

// Device-side association.  d_xyz [n][3]; the scan of every point either as d_frame [n] (0..W-1) or -- d_first != NULL -- as the offsets
// of the W scans in the point list (W + 1 entries, points scan by scan); d_poses [W][12],
// W = scans including the `fix_frames` marginalised ones.  `arena` / `arena_cap`: caller-owned scratch (may be NULL / 0);
// *arena_need receives the bytes this call wanted.  On success *F_out features; *d_out = hipMalloc'ed
// [F][W - fix_frames][10] (caller frees), *d_coe = [F], *d_fix = [F][10], *d_layer = [F], and, when want_points,
// *d_point_feat = [n] feature of every point (-1: none).  Returns 0, or a negative code (-1 allocation / HIP
// failure, -2 unsupported size, -3 a scan index outside [0, W) or a non-finite point).
int associate_device(hipStream_t s, const float *d_xyz, const int *d_frame, const long *d_first, const double *d_poses, long n, const AssocOpts &o,
                     void *arena, size_t arena_cap, size_t *arena_need, int *F_out, double **d_out, double **d_coe,
                     double **d_fix, int **d_layer, int **d_point_feat, long *n_roots, AssocMail *mail, bool *outputs_owned) {
  *F_out = 0; *outputs_owned = true; *d_out = nullptr; *d_coe = nullptr; *d_fix = nullptr; *d_layer = nullptr; *n_roots = 0;
  if (d_point_feat) *d_point_feat = nullptr;
  const int W = o.W;
  if (W > 512 || n <= 0 || n >= (1l << 31) || o.layer_limit < 0 || o.layer_limit > 2 || o.fix_frames < 0 || o.fix_frames >= W)
    return -2;
  const bool strict = o.max_dis > 0 || o.ratio21_max > 0 || o.lam0_max > 0;
  const bool want_points = d_point_feat != nullptr;
  const int levels = o.layer_limit + 1;
  if (!d_frame && !d_first) return -1;
  const ScanOf scan{d_first ? nullptr : d_frame, d_first, W};      // (d_first: the offsets of the W scans in the point list, W + 1 entries)
  Scratch sc(arena, arena_cap);
  sc.persist = mail;
  struct NeedOut { Scratch &sc; size_t *out; ~NeedOut() { *out = sc.need; } } need_out{sc, arena_need};
  const int B = 256;
  auto *k0 = sc.get<unsigned long long>(n), *k0s = sc.get<unsigned long long>(n);   // k0 is reused as the level key
  auto *val = sc.get<unsigned long long>(n), *vals = sc.get<unsigned long long>(n);
  auto *idx1 = sc.get<unsigned int>(n), *idxL = sc.get<unsigned int>(n);
  auto *rootid = sc.get<unsigned int>(n), *incl = sc.get<unsigned int>(n);
  auto *cks = sc.get<unsigned long long>(n);
  auto *range = sc.get<int>(RANGE_ROW * RANGE_BLOCKS);
  unsigned int *pnode[3] = {nullptr, nullptr, nullptr};
  if (want_points) for (int L = 0; L < levels; L++) pnode[L] = sc.get<unsigned int>(n);
  if (!sc.ok) return -1;

  // pass A: key range -> digits needed per axis
  const int rblocks = std::min(grid_for(n, B), RANGE_BLOCKS);
  std::vector<int> h_rows((size_t)RANGE_ROW * rblocks);
  hipLaunchKernelGGL(k_vox_range, dim3(rblocks), dim3(B), 0, s, d_xyz, scan, d_poses, n, W, o.voxel_size, range);
  hipMemcpyAsync(h_rows.data(), range, h_rows.size() * sizeof(int), hipMemcpyDeviceToHost, s);
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  int h_range[6] = {1 << 30, 1 << 30, 1 << 30, -(1 << 30), -(1 << 30), -(1 << 30)};
  bool scan_ordered = true;       // points arrive scan by scan (every driver of the reference): stable sorts keep that order
  for (int b = 0; b < rblocks; b++) {
    if (h_rows[(size_t)RANGE_ROW * b + 6] & 1) return -3;       // scan index out of range / non-finite point
    if (h_rows[(size_t)RANGE_ROW * b + 6] & 2) scan_ordered = false;
    for (int j = 0; j < 3; j++) {
      h_range[j] = std::min(h_range[j], h_rows[(size_t)RANGE_ROW * b + j]);
      h_range[3 + j] = std::max(h_range[3 + j], h_rows[(size_t)RANGE_ROW * b + 3 + j]);
    }
  }
  KeyPack kp;
  int key_bits = 1;
  {
    unsigned long long cells = 1;                 // < 2^66 / 8: three spans below 2^22 each
    for (int j = 0; j < 3; j++) {
      if (h_range[j] <= -(1 << 21) || h_range[3 + j] >= (1 << 21)) return -2;   // not a LiDAR map at this voxel size
      kp.off[j] = h_range[j];
      kp.n[j] = (unsigned long long)((long long)h_range[3 + j] - h_range[j]) + 1ull;
      if (cells > (~0ull) / kp.n[j]) return -2;
      cells *= kp.n[j];
    }
    while (key_bits < 64 && ((cells - 1) >> key_bits)) key_bits++;
  }

  VoxParams pr{o.voxel_size, {o.thr[0], o.thr[1], o.thr[2]}, o.min_ps, W, o.layer_limit, o.min_observers, o.fix_frames,
               o.max_dis, o.ratio21_max, o.lam0_max};
  Level lv[3];
  int fb = 1;                                     // bits of a scan index
  while ((1 << fb) < W) fb++;
  // Round 5's path (records in root order, levels 1-2 as stable partitions inside the roots: see k_part_scatter) whenever the points
  // arrive scan by scan and the packed root key fits 32 bits; BALM_ASSOC=sorted forces the library-sort path (A/B, tests)
  const char *amode = getenv("BALM_ASSOC");
  const bool fast_keys = scan_ordered && key_bits <= 32 && !(amode && !strcmp(amode, "sorted"));
  unsigned short *tag = nullptr, *tags = nullptr;
  unsigned int *idx0 = (unsigned int *)val, *idx0s = (unsigned int *)vals;       // (the fast path's sort values live in the u64 arrays)
  // round 6: the root order by multisplit (see k_ms_keys) when the key space is small enough for a bitmap and -- known after M2 -- the
  // window has few enough roots; BALM_ASSOC=radix keeps round 5's radix sort of (key, index) pairs
  const bool want_ms = fast_keys && key_bits <= 26 && !(amode && !strcmp(amode, "radix"));
  bool ms = false;
  long NR = -1;
  unsigned int *ms_bitmap = nullptr, *ms_rank = nullptr;
  const long ms_tiles = (n + MS_TILE - 1) / MS_TILE;
  if (fast_keys) {
    tag = sc.get<unsigned short>(n); tags = sc.get<unsigned short>(n);
    if (!sc.ok) return -1;
    if (want_ms) {
      const long nw = (long)(((1ull << key_bits) + 31ull) / 32ull);
      ms_bitmap = sc.get<unsigned int>(nw); ms_rank = sc.get<unsigned int>(nw + 1);
      if (!sc.ok) return -1;
      hipMemsetAsync(ms_bitmap, 0, (size_t)nw * sizeof(unsigned int), s);
      hipLaunchKernelGGL((k_ms_keys<unsigned int>), dim3(grid_for(n, B)), dim3(B), 0, s, d_xyz, scan, d_poses, n, o.voxel_size, kp,
                         (unsigned int *)k0, tag, ms_bitmap);
      {
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<long>(0), PopcOf{ms_bitmap, nw});
        size_t tmp = 0;
        rocprim::exclusive_scan(nullptr, tmp, in, ms_rank, 0u, (size_t)(nw + 1), rocprim::plus<unsigned int>(), s);
        void *d = sc.get<char>(tmp);
        if (!d) return -1;
        rocprim::exclusive_scan(d, tmp, in, ms_rank, 0u, (size_t)(nw + 1), rocprim::plus<unsigned int>(), s);
      }
      NR = last_u32(s, ms_rank, nw + 1, mail);
      int rb = 1;
      while ((1l << rb) < NR) rb++;
      ms = NR >= 1 && NR <= MS_MAX_ROOTS && rb + 3 * (levels - 1) + fb <= 32 && ms_tiles * NR <= (64l << 20);
    }
    if (!ms) {                     // round 5's root order (also: more roots than the multisplit's LDS tables hold)
      hipLaunchKernelGGL((k_vox_keys_tag<unsigned int>), dim3(grid_for(n, B)), dim3(B), 0, s, d_xyz, scan, d_poses, n, o.voxel_size, kp,
                         (unsigned int *)k0, idx0, tag);
      sort_pairs(sc, s, (unsigned int *)k0, (unsigned int *)k0s, idx0, idx0s, n, key_bits);
      scan_heads(sc, s, (const unsigned int *)k0s, 0, rootid, n);
      NR = -1;
    }
  } else {
    auto root_keys = [&](auto *ka, auto *kb) {     // 32-bit radix keys whenever the packed key fits
      using K = std::remove_pointer_t<decltype(ka)>;
      hipLaunchKernelGGL((k_vox_keys<K>), dim3(grid_for(n, B)), dim3(B), 0, s, d_xyz, scan, d_poses, n, o.voxel_size, kp, ka, val);
      sort_pairs(sc, s, ka, kb, val, vals, n, key_bits);
      scan_heads(sc, s, (const K *)kb, 0, rootid, n);
    };
    if (key_bits <= 32) root_keys((unsigned int *)k0, (unsigned int *)k0s);
    else root_keys(k0, k0s);
  }
  if (!sc.ok) return -1;
  if (NR < 0) NR = last_u32(s, rootid, n, mail);
  int root_bits = 1;
  while ((1l << root_bits) < NR) root_bits++;
  const bool fast = fast_keys && root_bits + 3 * (levels - 1) + fb <= 32;

  // everything of a level behind its sorted list: cks = the composite keys in list order (32- or 64-bit), idxL = the points' indices
  // in list order (NULL when nobody needs them), recL = the records in list order (fast path) or NULL (gather through idxL)
  // nL = entries of the level's list: all n points, or (fast path, levels 1 and 2) the points of the split roots only
  // Three phases, so that levels whose lists exist together (the fast path's levels 1 and 2) can share the two counts the host has to
  // read per level -- segments, then nodes -- in ONE mailbox round trip each:
  //   level_scan      segment ranks of the list's entries (inclusive scan of the head flags) -> st.incl; the count is its last entry
  //   level_segments  the segments' keys, starts and clusters; node ranks of the segments -> st.nid (the node count is its last entry)
  //   level_nodes     node tables, totals, plane tests
  struct LevelStage {
    long nL = 0; bool narrow = true; const void *cks = nullptr; const unsigned int *idx = nullptr; const uint4 *rec = nullptr;
    unsigned int *incl = nullptr, *seg_start = nullptr, *nid = nullptr, *pid = nullptr; double *seg_world = nullptr;
  } stage[3];
  auto level_scan = [&](int L, long nL, bool narrow, const void *cks_, const unsigned int *idxL_, const uint4 *recL, unsigned int *incl_) -> int {
    LevelStage &st = stage[L];
    st.nL = nL; st.narrow = narrow; st.cks = cks_; st.idx = idxL_; st.rec = recL; st.incl = incl_;
    // (rocprim::select on the head flags -- the run starts compacted in one pass, no ranks -- was measured too: 147 us per level on the
    //  shipped window against ~100 for this scan + k_seg_heads)
    if (narrow) scan_heads(sc, s, (const unsigned int *)cks_, 0, st.incl, nL);
    else scan_heads(sc, s, (const unsigned long long *)cks_, 0, st.incl, nL);
    return sc.ok ? 0 : -1;
  };
  auto level_segments = [&](int L, long NS) -> int {
    Level &v = lv[L];
    LevelStage &st = stage[L];
    const long nL = st.nL;
    v.NS = NS;
    st.seg_start = sc.get<unsigned int>(v.NS + 1);
    v.seg_ck = sc.get<unsigned long long>(v.NS);
    v.seg_body = sc.get<double>((size_t)v.NS * 10);
    st.seg_world = sc.get<double>((size_t)v.NS * 10);
    st.nid = sc.get<unsigned int>(v.NS); st.pid = sc.get<unsigned int>(v.NS);
    v.seg_node = st.nid;
    unsigned int *lists = st.rec ? sc.get<unsigned int>(4 + 8 * (size_t)v.NS) : nullptr;
    if (!sc.ok) return -1;
    if (st.narrow)
      hipLaunchKernelGGL((k_seg_heads<unsigned int>), dim3(grid_for(nL, B)), dim3(B), 0, s, (const unsigned int *)st.cks, st.incl, nL, L,
                         fb, st.seg_start, v.seg_ck, lists);
    else
      hipLaunchKernelGGL((k_seg_heads<unsigned long long>), dim3(grid_for(nL, B)), dim3(B), 0, s, (const unsigned long long *)st.cks, st.incl, nL, L, fb,
                         st.seg_start, v.seg_ck, lists);
    if (st.rec) {
      launch_seg_clusters_rec(s, st.rec, d_poses, st.seg_start, v.seg_ck, v.NS, v.seg_body, st.seg_world, lists);
    }
    else launch_seg_clusters(s, d_xyz, d_poses, st.idx, st.seg_start, v.seg_ck, v.NS, v.seg_body, st.seg_world);
    scan_heads(sc, s, (const unsigned long long *)v.seg_ck, 9, st.nid, v.NS);
    // a node's parent: level 1 -> its root voxel, whose index the key carries (the compact level lists hold the split roots only:
    // a rank among THEM is not a root index); level 2 -> the level-1 node = the rank of its (root, octant) prefix, which both lists share
    if (L == 2) scan_heads(sc, s, (const unsigned long long *)v.seg_ck, 12, st.pid, v.NS);
    return sc.ok ? 0 : -1;
  };
  auto level_nodes = [&](int L, long NN) -> int {
    Level &v = lv[L];
    LevelStage &st = stage[L];
    const int pshift = L == 2 ? 12 : 0;
    v.NN = NN;
    v.node_seg = sc.get<unsigned int>(v.NN + 1);
    v.node_parent = sc.get<unsigned int>(v.NN);
    v.tot = sc.get<NodeTot>(v.NN);
    v.status = sc.get<unsigned char>(v.NN);
    double *plane = strict ? sc.get<double>((size_t)v.NN * 6) : nullptr;
    if (!sc.ok) return -1;
    hipLaunchKernelGGL(k_level_heads, dim3(grid_for(v.NS, B)), dim3(B), 0, s, v.seg_ck, st.nid, st.pid, v.NS, pshift, v.node_seg,
                       v.node_parent);
    hipLaunchKernelGGL(k_node_totals_status, dim3(grid_for(v.NN * 16, B)), dim3(B), 0, s, st.seg_world, v.seg_ck, v.node_seg, v.NN,
                       o.fix_frames, v.tot, pr.thr[L], pr, v.status, plane);
    if (strict && o.max_dis > 0)
      hipLaunchKernelGGL(k_point_plane_dist, dim3(grid_for(st.nL, B)), dim3(B), 0, s, d_xyz, scan, d_poses, st.idx, st.incl, st.nid, st.nL, plane,
                         o.max_dis, v.status);
    if (want_points) hipLaunchKernelGGL(k_point_nodes, dim3(grid_for(st.nL, B)), dim3(B), 0, s, st.idx, st.incl, st.nid, st.nL, pnode[L]);
    return 0;           // (flag[NN] = 0: k_feature_flags)
  };
  // one level on its own: scan, count, segments, count, nodes (nn_known > 0: the node count is known -- level 0's nodes ARE the roots)
  auto finish_level = [&](int L, long nL, bool narrow, const void *cks_, const unsigned int *idxL_, const uint4 *recL, long nn_known = 0) -> int {
    if (level_scan(L, nL, narrow, cks_, idxL_, recL, incl)) return -1;
    if (level_segments(L, last_u32(s, incl, nL, mail))) return -1;
    return level_nodes(L, nn_known > 0 ? nn_known : (long)last_u32(s, stage[L].nid, lv[L].NS, mail));
  };

  if (fast) {
    const bool need_idx = want_points || (strict && o.max_dis > 0);       // who still wants the points' original indices per level
    auto *rec0 = sc.get<uint4>(n);
    auto *ck0 = (unsigned int *)cks;
    auto *root_start = sc.get<unsigned int>(NR + 1), *live_start = sc.get<unsigned int>(NR + 1), *tile_base = sc.get<unsigned int>(NR + 1);
    auto *plan = sc.get<unsigned int>(4);
    if (!sc.ok) return -1;
    if (ms) {
      auto *rid = sc.get<unsigned short>(n);
      auto *hist = sc.get<unsigned int>((size_t)ms_tiles * NR), *gsum = sc.get<unsigned int>((size_t)MS_GROUPS * NR);
      if (!sc.ok) return -1;
      const int per_group = (int)((ms_tiles + MS_GROUPS - 1) / MS_GROUPS), groups = (int)((ms_tiles + per_group - 1) / per_group);
      const dim3 rgrid((unsigned int)((NR + 255) / 256), (unsigned int)groups);
      hipLaunchKernelGGL(k_ms_hist, dim3((unsigned int)ms_tiles), dim3(256), 0, s, (const unsigned int *)k0, n, ms_bitmap, ms_rank, (int)NR, rid, hist);
      hipLaunchKernelGGL(k_ms_group_sums, rgrid, dim3(256), 0, s, hist, (int)NR, ms_tiles, per_group, gsum);
      hipLaunchKernelGGL(k_ms_bases, dim3(1), dim3(1024), 0, s, gsum, (int)NR, groups, n, root_start);
      hipLaunchKernelGGL(k_ms_tile_offsets, rgrid, dim3(256), 0, s, hist, (int)NR, ms_tiles, per_group, gsum);
      hipLaunchKernelGGL(k_ms_scatter, dim3((unsigned int)ms_tiles), dim3(256), 0, s, d_xyz, tag, rid, n, hist, (int)NR, root_bits, fb, rec0, ck0, tags,
                         need_idx ? idx0s : (unsigned int *)nullptr);
    } else {
      hipLaunchKernelGGL(k_gather_records, dim3(grid_for(n, B)), dim3(B), 0, s, d_xyz, tag, idx0s, rootid, n, fb, rec0, ck0, tags);
      if (levels > 1) hipLaunchKernelGGL(k_root_starts, dim3(grid_for(n, B)), dim3(B), 0, s, (const unsigned int *)k0s, rootid, n, NR, root_start);
    }
    if (finish_level(0, n, true, ck0, idx0s, rec0, NR)) return -1;          // (ck0 numbers the roots 0 .. NR - 1: level 0 has NR nodes)
    if (levels > 1) {
      // levels 1 and 2 for the points of the roots recut splits (status NODE_SPLIT, known now), in compact lists
      if (want_points) for (int L = 1; L < levels; L++) hipMemsetAsync(pnode[L], 0xff, (size_t)n * sizeof(unsigned int), s);
      hipLaunchKernelGGL(k_live_plan, dim3(1), dim3(1024), 0, s, root_start, lv[0].status, NR, tile_base, live_start, plan);
      const unsigned int *src[4] = {plan, plan + 1, plan, plan};
      unsigned int got[4] = {0, 0, 0, 0};
      if (!mail_u32x4(s, mail, src, got)) return -1;
      const long NT = got[0], n_live = got[1];
      if (n_live > 0) {
        auto *hist = sc.get<unsigned int>((size_t)NT * 64), *off1 = sc.get<unsigned int>((size_t)NT * 8), *off2 = sc.get<unsigned int>((size_t)NT * 64);
        uint4 *rec1 = sc.get<uint4>(n_live), *rec2 = nullptr;
        unsigned int *ck1 = sc.get<unsigned int>(n_live), *ck2 = nullptr, *idx1 = nullptr, *idx2 = nullptr;
        if (levels > 2) { rec2 = sc.get<uint4>(n_live); ck2 = sc.get<unsigned int>(n_live); }
        if (need_idx) { idx1 = sc.get<unsigned int>(n_live); if (levels > 2) idx2 = sc.get<unsigned int>(n_live); }
        if (!sc.ok) return -1;
        hipLaunchKernelGGL(k_part_hist, dim3((unsigned int)NT), dim3(256), 0, s, tags, root_start, tile_base, (int)NR, hist);
        hipLaunchKernelGGL(k_part_offsets, dim3((unsigned int)NR), dim3(64), 0, s, hist, live_start, tile_base, (int)NR, off1, off2);
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned int)NT), dim3(256), 0, s, rec0, idx0s, root_start, tile_base, (int)NR, off1, off2, fb,
                           levels, rec1, ck1, idx1, rec2, ck2, idx2);
        if (levels == 2) {
          if (finish_level(1, n_live, true, ck1, idx1, rec1)) return -1;
        } else {                                   // both lists exist: their counts travel together
          auto *incl2 = sc.get<unsigned int>(n_live);
          if (!sc.ok) return -1;
          if (level_scan(1, n_live, true, ck1, idx1, rec1, incl) || level_scan(2, n_live, true, ck2, idx2, rec2, incl2)) return -1;
          const unsigned int *ns_src[4] = {incl + n_live - 1, incl2 + n_live - 1, incl, incl};
          unsigned int ns[4] = {0, 0, 0, 0};
          if (!mail_u32x4(s, mail, ns_src, ns)) return -1;
          if (level_segments(1, ns[0]) || level_segments(2, ns[1])) return -1;
          const unsigned int *nn_src[4] = {stage[1].nid + ns[0] - 1, stage[2].nid + ns[1] - 1, incl, incl};
          unsigned int nn[4] = {0, 0, 0, 0};
          if (!mail_u32x4(s, mail, nn_src, nn)) return -1;
          if (level_nodes(1, nn[0]) || level_nodes(2, nn[1])) return -1;
        }
      }
    }
  } else {
    if (fast_keys) {          // the root sort ran on (key, index): the sorted path's 64-bit values from the tags
      hipLaunchKernelGGL(k_vals_from_tags, dim3(grid_for(n, B)), dim3(B), 0, s, idx0s, tag, n, k0);
      hipMemcpyAsync(vals, k0, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
    }
    for (int L = 0; L < levels; L++) {
      const int key_bits_L = root_bits + 3 * L + fb;
      // The keys are built in root-sorted order, which is scan order inside every root voxel when the points arrived scan by
      // scan (the root sort is stable).  Then (i) the level-0 key (root, scan) is already sorted: no sort at all; (ii) deeper
      // levels only need the bits ABOVE the scan index sorted -- the stable passes keep the scans in order for free:
      // 14 / 17 bits instead of 22 / 25 on the shipped window = 5 radix passes instead of 10 over the three levels.
      auto level_keys = [&](auto *ka, auto *kb) {
        using K = std::remove_pointer_t<decltype(ka)>;
        if (scan_ordered && L == 0) {
          hipLaunchKernelGGL((k_make_ck<K>), dim3(grid_for(n, B)), dim3(B), 0, s, rootid, vals, n, L, fb, kb, idxL);
        } else {
          hipLaunchKernelGGL((k_make_ck<K>), dim3(grid_for(n, B)), dim3(B), 0, s, rootid, vals, n, L, fb, ka, idx1);
          sort_pairs(sc, s, ka, kb, idx1, idxL, n, key_bits_L, scan_ordered ? fb : 0);
        }
      };
      const bool narrow = key_bits_L <= 32;
      if (narrow) level_keys((unsigned int *)k0, (unsigned int *)cks);
      else level_keys(k0, cks);
      if (finish_level(L, n, narrow, cks, idxL, nullptr)) return -1;
    }
  }
  if (lv[0].NN != NR) return -1;
  unsigned int FL[3] = {0, 0, 0};
  {
    // every level's nodes (a level without nodes -- no root was split -- is an empty stretch) + its zero entry, end to end
    FlagLevels fl;
    long total = 0;
    for (int L = 0; L < 3; L++) {
      fl.NN[L] = L < levels ? lv[L].NN : 0; fl.tot[L] = lv[L].tot; fl.off[L] = total;
      if (L < levels) total += fl.NN[L] + 1;
    }
    auto *flag_all = sc.get<unsigned int>(total), *fid_all = sc.get<unsigned int>(total);
    if (!sc.ok) return -1;
    for (int L = 0; L < levels; L++) { lv[L].flag = flag_all + fl.off[L]; lv[L].fid = fid_all + fl.off[L]; }
    hipLaunchKernelGGL(k_feature_flags, dim3(grid_for(total, B)), dim3(B), 0, s, fl, levels, lv[0].status, lv[1].status, lv[2].status,
                       lv[1].node_parent, lv[2].node_parent, pr, flag_all);
    scan_excl(sc, s, flag_all, fid_all, total);
    if (!sc.ok) return -1;
    const unsigned int *src[4];
    for (int k = 0; k < 4; k++) src[k] = fid_all + fl.off[k < levels ? k : levels - 1] + fl.NN[k < levels ? k : levels - 1];     // the zero entries: running counts
    unsigned int got[4] = {0, 0, 0, 0};
    if (!mail_u32x4(s, mail, src, got)) return -1;
    for (int L = 0; L < levels; L++) FL[L] = got[L] - (L ? got[L - 1] : 0u);
  }
  const long F = (long)FL[0] + FL[1] + FL[2];
  *n_roots = NR;
  if (hipGetLastError() != hipSuccess) return -1;
  if (F == 0) return 0;
  const int Wout = W - o.fix_frames;
  double *out = nullptr, *coe = nullptr, *fixo = nullptr;
  int *lay = nullptr, *pf = nullptr;
  // the feature table the caller installs right away: out of the arena when it fits (no hipMalloc / hipFree per call), else the
  // caller's to free
  const size_t out_bytes = ((size_t)F * Wout * 10 + (size_t)F + (size_t)F * 10) * sizeof(double) + (size_t)F * sizeof(int) +
                           (want_points ? (size_t)n * sizeof(int) : 0) + 5 * 256;
  const bool from_arena = sc.off + out_bytes <= sc.cap;
  auto fail = [&]() { if (!from_arena) { if (out) hipFree(out); if (coe) hipFree(coe); if (fixo) hipFree(fixo); if (lay) hipFree(lay); if (pf) hipFree(pf); } return -1; };
  if (from_arena) {
    out = sc.get<double>((size_t)F * Wout * 10); coe = sc.get<double>((size_t)F); fixo = sc.get<double>((size_t)F * 10);
    lay = sc.get<int>((size_t)F); if (want_points) pf = sc.get<int>((size_t)n);
    *outputs_owned = false;
  } else {
    sc.need += out_bytes;
    if (hipMalloc((void **)&out, (size_t)F * Wout * 10 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&coe, (size_t)F * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&fixo, (size_t)F * 10 * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&lay, (size_t)F * sizeof(int)) != hipSuccess ||
        (want_points && hipMalloc((void **)&pf, (size_t)n * sizeof(int)) != hipSuccess))
      return fail();
  }
  hipMemsetAsync(out, 0, (size_t)F * Wout * 10 * sizeof(double), s);
  for (int L = 0; L < levels; L++) {
    if (lv[L].NN == 0 || FL[L] == 0) continue;
    const int seg_blocks = grid_for(lv[L].NS * 16, B);
    hipLaunchKernelGGL(k_emit_level, dim3(seg_blocks + grid_for(lv[L].NN * 16, B)), dim3(B), 0, s, lv[L].NS, seg_blocks, lv[L].seg_node, lv[L].flag,
                       lv[L].fid, lv[L].seg_ck, lv[L].seg_body, Wout, o.fix_frames, out, lv[L].NN, lv[L].tot, L, coe, fixo, lay);
  }
  if (want_points)         // (the scanned values are indices in the whole table: no per-level base)
    hipLaunchKernelGGL(k_point_features, dim3(grid_for(n, B)), dim3(B), 0, s, n, levels, pnode[0], pnode[1], pnode[2], lv[0].flag,
                       lv[1].NN ? lv[1].flag : (unsigned int *)nullptr, lv[2].NN ? lv[2].flag : (unsigned int *)nullptr, lv[0].fid, lv[1].fid, lv[2].fid, 0u, 0u, pf);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return fail();
  *F_out = (int)F; *d_out = out; *d_coe = coe; *d_fix = fixo; *d_layer = lay;
  if (want_points) *d_point_feat = pf;
  return 0;
}

#include "kernels_window.inc"

// the code object of this translation unit, loaded on the current device now (the runtime loads it on the first use of any of its
// kernels otherwise: balm_prewarm does it on a background thread while the caller is still busy elsewhere)
hipError_t preload_voxel() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_set_u32);
}

}  // namespace balm
