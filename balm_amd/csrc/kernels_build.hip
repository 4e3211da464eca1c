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

// TERMS = false: one lane per run pushes all ten sums of its run (short runs: 64 runs fill the wave).
// TERMS = true : one lane per (run, term column) -- the nine sums of a run are nine independent in-order chains, so a block
//                of few long runs (the launch default: 40 points per (feature, pose), 13 runs per 512-point block) still
//                occupies 117 lanes instead of 13, and a lane's chain is one rounded product + one rounded add per point
//                (tools.hpp:311-316: P += v v^T entry by entry, v += p; the same k_seg_clusters_long uses in kernels_voxel.hip).
//                Order and rounding per sum are untouched: bit-identical to PointCluster::push either way.
// The run that crosses the end of a block belongs to the wave that holds its head.  Its continuation used to be walked
// point by point from memory by one lane (a dependent ~0.8 us round trip per point: 16 us per block at 40-point runs, 65 %
// of the kernel); now the 64 points behind the block are loaded with the block's own loads (one more coalesced load per
// lane, no extra round trip), the run's end is found by ballot, and only a run that goes on for more than those 64 points
// takes the chunked path: 64 points per coalesced load, the next chunk in flight while the current one is added.
template <int BUILD_BP, bool TERMS>
__global__ __launch_bounds__(256) void k_build_clusters_runs(const float *__restrict__ xyz, const int *__restrict__ fid,
                                                             const int *__restrict__ pid, long n_pts, int F, int W,
                                                             double *__restrict__ soa, int *__restrict__ unsorted) {
  __shared__ f3 pts[4][BUILD_BP + 64];
  __shared__ short rstart[4][BUILD_BP + 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long nblk = (n_pts + BUILD_BP - 1) / BUILD_BP;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
  const long key_end = (long)F * W;                       // past every valid key: lanes beyond the last point
  const f3 *__restrict__ P3 = reinterpret_cast<const f3 *>(xyz);
  auto key_of = [&](int a, int i) -> long { return (a >= 0 && a < F && i >= 0 && i < W) ? (long)a * W + i : -1; };
  const float *pf = &pts[wv][0].x;                        // pts as a flat float array: component c of point l = pf[3 l + c]
  // (Measured and rejected, round 3: persistent waves sized to the resident grid with the NEXT block's loads in flight while
  // the current block is worked on -- 0.180 instead of 0.170 ms at 6-point runs, 0.169 instead of 0.142 at 40-point runs: the
  // second register set costs a wave per SIMD, and with ~10 blocks per wave the stride leaves the last round half empty.)
  constexpr int NJ = BUILD_BP / 64;
  for (long blk = wave; blk < nblk; blk += nwave) {
    const long t0 = blk * BUILD_BP;
    const int nloc = (int)(n_pts - t0 < BUILD_BP ? n_pts - t0 : BUILD_BP);
    // ---- every load of the block is issued before the first use (the kernel is latency-bound otherwise), the 64 points
    // behind the block included (the continuation of the block's last run)
    f3 q[NJ + 1]; int fa[NJ + 1], fi[NJ + 1];
    long carry = -2;
    if (t0 > 0) carry = key_of(fid[t0 - 1], pid[t0 - 1]);  // key of the point before this block (wave-uniform)
#pragma unroll
    for (int j = 0; j <= NJ; j++) {
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
    // ---- the last run's continuation behind the block: m leading points of the next 64 carry the block's last key
    int tail = 0;
    if (nloc == BUILD_BP) {
      const long t = t0 + BUILD_BP + lane;
      const long key = t < n_pts ? key_of(fa[NJ], fi[NJ]) : key_end;
      const unsigned long long differ = __ballot(key != carry);
      tail = differ ? __builtin_ctzll(differ) : 64;
      pts[wv][BUILD_BP + lane] = q[NJ];
    }
    if (lane == 0) rstart[wv][runs] = (short)(nloc + tail);
    const bool more = tail == 64;                         // the last run goes on beyond the staged points
    const long last_key = carry;
    if (!TERMS) {
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
        for (int l = s0; l < s1; l++) push(pts[wv][l]);       // (unrolled by four: slower, 0.171 vs 0.142 ms at 40-point runs)
        if (more && r == runs - 1)      // a run more than 64 points longer than its block (rare with short runs): from memory
          for (long tt = t0 + BUILD_BP + 64; tt < n_pts && key_of(fid[tt], pid[tt]) == key; tt++) push(P3[tt]);
        if (key >= 0) {      // ... and the wave that owns the next block finds no head at its first point
          double *dst = soa + (size_t)ai.x * 10 * W + ai.y;
          dst[0] = pxx; dst[(size_t)W] = pxy; dst[(size_t)2 * W] = pxz; dst[(size_t)3 * W] = pyy; dst[(size_t)4 * W] = pyz;
          dst[(size_t)5 * W] = pzz; dst[(size_t)6 * W] = vx; dst[(size_t)7 * W] = vy; dst[(size_t)8 * W] = vz; dst[(size_t)9 * W] = cnt;
        }
      }
    } else {
      // ---- one lane per (term column, run): column c of run r is the in-order sum of round(x_i * x_j) (c < 6) or of the
      // coordinate (c >= 6; as x * 1.0, exact).  Task u = c * nr + r: lanes of one column are neighbours, so that their
      // stores to consecutive poses are contiguous.  A last run that goes on beyond the staged points is left to the
      // chunked phase below.
      const int nr = more ? runs - 1 : runs;
      for (int u = lane; u < 9 * nr; u += 64) {
        const int c = u / nr, r = u - c * nr;
        const int s0 = rstart[wv][r], s1 = rstart[wv][r + 1];
        const int ci = c < 3 ? 0 : (c < 5 ? 1 : (c == 5 ? 2 : c - 6)), cj = c < 3 ? c : (c < 5 ? c - 2 : 2);
        double acc = 0.0;
        const double one = c < 6 ? 0.0 : 1.0;              // coordinate columns: x * 1.0 (exact), so that every lane runs one loop
        int l = s0;
        for (; l + 4 <= s1; l += 4) {                      // four points' operands in flight per LDS round trip, added in order
          const float a0 = pf[3 * l + ci], a1 = pf[3 * l + 3 + ci], a2 = pf[3 * l + 6 + ci], a3 = pf[3 * l + 9 + ci];
          const float b0 = pf[3 * l + cj], b1 = pf[3 * l + 3 + cj], b2 = pf[3 * l + 6 + cj], b3 = pf[3 * l + 9 + cj];
          acc = __dadd_rn(acc, __dmul_rn((double)a0, c < 6 ? (double)b0 : one));
          acc = __dadd_rn(acc, __dmul_rn((double)a1, c < 6 ? (double)b1 : one));
          acc = __dadd_rn(acc, __dmul_rn((double)a2, c < 6 ? (double)b2 : one));
          acc = __dadd_rn(acc, __dmul_rn((double)a3, c < 6 ? (double)b3 : one));
        }
        for (; l < s1; l++) acc = __dadd_rn(acc, __dmul_rn((double)pf[3 * l + ci], c < 6 ? (double)pf[3 * l + cj] : one));
        const int a = fid[t0 + s0], i = pid[t0 + s0];
        if (key_of(a, i) >= 0) {
          double *dst = soa + (size_t)a * 10 * W + i;
          dst[(size_t)c * W] = acc;
          if (c == 0) dst[(size_t)9 * W] = (double)(s1 - s0);       // N: a sum of ones, exact
        }
      }
      if (more && runs > 0) {
        // the long run: lanes 0..8 own its nine columns through the staged points and then through chunks of 64 points,
        // each one coalesced load per lane, the next chunk in flight while this one is added
        const int s0 = rstart[wv][runs - 1];
        const int c = lane < 9 ? lane : 0;
        const int ci = c < 3 ? 0 : (c < 5 ? 1 : (c == 5 ? 2 : c - 6)), cj = c < 3 ? c : (c < 5 ? c - 2 : 2);
        double acc = 0.0, cnt = (double)(BUILD_BP + 64 - s0);
        auto add_range = [&](int l0, int l1) {
          if (c < 6) for (int l = l0; l < l1; l++) acc = __dadd_rn(acc, __dmul_rn((double)pf[3 * l + ci], (double)pf[3 * l + cj]));
          else for (int l = l0; l < l1; l++) acc = __dadd_rn(acc, (double)pf[3 * l + ci]);
        };
        add_range(s0, BUILD_BP + 64);
        long tb = t0 + BUILD_BP + 64;
        f3 qn = f3{0.f, 0.f, 0.f}; long kn = key_end;
        if (tb + lane < n_pts) { qn = P3[tb + lane]; kn = key_of(fid[tb + lane], pid[tb + lane]); }
        for (;;) {
          const f3 qc = qn; const long kc = kn;
          const long tnext = tb + 64;
          qn = f3{0.f, 0.f, 0.f}; kn = key_end;
          if (tnext + lane < n_pts) { qn = P3[tnext + lane]; kn = key_of(fid[tnext + lane], pid[tnext + lane]); }
          const unsigned long long differ = __ballot(kc != last_key);
          const int m = differ ? __builtin_ctzll(differ) : 64;
          pts[wv][lane] = qc;                          // the block's own points are done with
          add_range(0, m);
          cnt += (double)m;
          if (m < 64) break;
          tb = tnext;
        }
        const int a = fid[t0 + s0], i = pid[t0 + s0];
        if (lane < 9 && key_of(a, i) >= 0) {
          double *dst = soa + (size_t)a * 10 * W + i;
          dst[(size_t)c * W] = acc;
          if (c == 0) dst[(size_t)9 * W] = cnt;
        }
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
  // long runs (the launch default is 40 points per (feature, pose)): one lane per (run, term column) instead of one per run,
  // in blocks of 256 points (6.4 runs x 9 columns = 58 of 64 lanes in ONE round, 88 VGPRs -> five waves per SIMD).  Measured
  // (profiles/r03c_cluster_build_ab.txt, 24 M points in 40-point runs): 0.125 ms = 0.53 of 8 TB/s; blocks of 512: 0.151;
  // one lane per run: 0.142 (512) / 0.226 (256).
  const char *em = getenv("BALM_BUILD_TERMS");           // A/B runs: 0 / 1 force
  const double avg_len = (double)n_pts / ((double)F * W > 1 ? (double)F * W : 1.0);
  const bool terms = em ? atoi(em) != 0 : avg_len >= 24.0;
  const char *e = getenv("BALM_BUILD_BP");               // A/B runs: 256 / 320 / 384 / 448 / 512
  int bp = e ? atoi(e) : 0;
  if (!bp && terms) bp = 256;
  if (!bp) {
    const double len = avg_len;
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
  do {                                                                                                                    \
    if (terms) hipLaunchKernelGGL((k_build_clusters_runs<BP, true>), dim3((unsigned)blocks), dim3(256), 0, s, xyz, feat_id, pose_id, n_pts, F, W, soa, flag); \
    else hipLaunchKernelGGL((k_build_clusters_runs<BP, false>), dim3((unsigned)blocks), dim3(256), 0, s, xyz, feat_id, pose_id, n_pts, F, W, soa, flag); \
  } while (0)
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

// The device's side of the strided point entries (host_stage.h StridedPoints): the caller's containers arrive as packed records and
// per-container COUNTS; the per-point container index (scan of balm_associate_scans, plane of balm_build_clusters_planes) is expanded
// here at HBM speed instead of crossing PCIe as one int per point (53.6 MB of the shipped window's 214 MB upload, VERDICT r5 Weak 7).
// first[k] = points before container k (k = 0..m, ascending, empty containers allowed): id[p] = the k with first[k] <= p < first[k+1].
__global__ __launch_bounds__(256) void k_expand_ids(const long *__restrict__ first, int m, long n, int *__restrict__ id) {
  extern __shared__ long sh_first[];
  const bool in_lds = m + 1 <= 4096;
  if (in_lds) {
    for (int t = threadIdx.x; t <= m; t += blockDim.x) sh_first[t] = first[t];
    __syncthreads();
  }
  const long *f = in_lds ? sh_first : first;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = m;                       // invariant: f[lo] <= p < f[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (f[mid] <= p) lo = mid; else hi = mid;
    }
    id[p] = lo;
  }
}

void launch_expand_ids(hipStream_t s, const long *d_first, int m, long n, int *d_id) {
  if (n <= 0 || m <= 0) return;
  long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  const size_t lds = m + 1 <= 4096 ? (size_t)(m + 1) * sizeof(long) : 0;
  hipLaunchKernelGGL(k_expand_ids, dim3((unsigned)blocks), dim3(256), lds, s, d_first, m, n, d_id);
}

__global__ __launch_bounds__(256) void k_rebase_ids(int *__restrict__ id, long n, int base) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) id[p] -= base;
}
void launch_rebase_ids(hipStream_t s, int *d_id, long n, int base) {
  if (n <= 0 || base == 0) return;
  long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rebase_ids, dim3((unsigned)blocks), dim3(256), 0, s, d_id, n, base);
}

// 16-byte records (x, y, z, w) -> packed xyz + (int)w: the (int)ap.intensity of benchmark_virtual.cpp:396, C truncation
__global__ __launch_bounds__(256) void k_unpack_xyzw(const float4 *__restrict__ rec, long n, float *__restrict__ xyz, int *__restrict__ aux) {
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
    const float4 r = rec[p];
    xyz[3 * p] = r.x; xyz[3 * p + 1] = r.y; xyz[3 * p + 2] = r.z;
    aux[p] = (int)r.w;
  }
}

void launch_unpack_xyzw(hipStream_t s, const float *d_rec, long n, float *d_xyz, int *d_aux) {
  if (n <= 0) return;
  long blocks = (n + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_unpack_xyzw, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4 *>(d_rec), n, d_xyz, d_aux);
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

// the code object of this translation unit, loaded on the current device now (the runtime loads it on the first use of any of its
// kernels otherwise: balm_prewarm does it on a background thread while the caller is still busy elsewhere)
hipError_t preload_build() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_unpack_xyzw);
}

}  // namespace balm
