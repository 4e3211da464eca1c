// Damped LM solve on gfx950:  (H + u diag H) dx = -g ,  q1 = 0.5 dx.(u D dx - g)
//   reference: src/benchmark/bavoxel.hpp:1113-1114,1127  (Eigen `.ldlt().solve()`)
//
// Eigen's LDLT (in-place lower, unblocked) searches its pivot on the NOT-yet-updated trailing
// diagonal (its update is left-looking), i.e. the elimination order is simply "decreasing
// |diagonal| of the input matrix".  We therefore apply that static symmetric permutation once and
// run an un-pivoted *blocked* LDL^T (D diagonal, possibly negative: the exact second-order Hessian
// is indefinite away from the optimum, SURVEY.md finding 4):
//   per panel of NB=48 columns:  ldl_panel (rank-4 steps on f64 MFMA) L11, D11 (redundantly per workgroup),
//                                                                     W21 = L21 D11, L21
//                                ldl_trail (f64 MFMA, 48x16 tiles)    A22 -= L21 W21^T
//   Two things ride along as extra rows below the matrix (ldA = 2 nA + NB rows of storage):
//     row nA          the right-hand side: the panel / trail kernels produce z = D^+ L^-1 P b for free;
//     rows nA+NB ..   the identity: its rows come out as M = L^-T D^+ (upper triangular; tile t is only touched
//                     from panel t on, so the factorisation carries ~25 % more trail tiles and no extra
//                     launches), which turns the backward solve  L^T x = z  -- a chain of one launch per panel
//                     -- into ONE matrix-vector product  x = M (D z)  (k_ldl_apply).
// Pose update kernels (bavoxel.hpp:1116-1126, 1159-1164) live here too.
#include <cfloat>

#include "balm_internal.h"

namespace balm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// permutation by decreasing |diag H| (ties by index); padded positions (>= n) go last.
// 16 ranks per workgroup, the j-range split sixteen ways (the kernel is on the critical path of every solve and
// has no other parallelism to offer: nA / 16 workgroups instead of nA / 64).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank_diag(const double *__restrict__ H, int n, int nA,
                                                   int *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) double dabs[];   // [nA] then int part[256]
  int *part = reinterpret_cast<int *>(dabs + nA);
  // a NaN diagonal ranks after every real entry and before the padding, so that perm stays a permutation
  for (int i = threadIdx.x; i < nA; i += blockDim.x) {
    const double v = i < n ? fabs(H[(size_t)i * n + i]) : -1.0;
    dabs[i] = v == v ? v : -0.5;
  }
  __syncthreads();
  const int il = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;
  int rank = 0;
  if (i < nA) {
    const double di = dabs[i];
    const int chunk = (nA + 15) / 16;
    const int j0 = q * chunk, j1 = min(nA, j0 + chunk);
    for (int j = j0; j < j1; j++) {
      const double dj = dabs[j];
      rank += (dj > di) || (dj == di && j < i);
    }
  }
  part[threadIdx.x] = rank;
  __syncthreads();
  if (q == 0 && i < nA) {
    rank = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) rank += part[16 * k + il];
    perm[rank] = i;
  }
}

__global__ __launch_bounds__(256) void k_build_A(const double *__restrict__ H, const double *__restrict__ g, int n,
                                                 int nA, const int *__restrict__ perm, double u,
                                                 double *__restrict__ A) {
  const int ldA = 2 * nA + NB;
  const long total = (long)ldA * nA;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / ldA), r = (int)(t - (long)c * ldA);
    const int pc = perm[c];
    double v;
    if (r < nA) {
      const int pr = perm[r];
      if (pr < n && pc < n) {
        v = H[(size_t)pc * n + pr];
        if (r == c) v += u * v;      // D = diag(H)  (bavoxel.hpp:1113)
      } else {
        v = (r == c) ? 1.0 : 0.0;
      }
    } else if (r < nA + NB) {
      v = (r == nA && pc < n) ? -g[pc] : 0.0;     // right-hand side row: P (-JacT)
    } else {
      v = (r - (nA + NB) == c) ? 1.0 : 0.0;       // identity rows -> L^-T D^+
    }
    A[t] = v;
  }
}

// wave-uniform broadcast of lane `src`'s value (src is a compile-time constant after unrolling)
__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// ldl_panel: LDL^T of the NB x NB diagonal block AND of this workgroup's 64 rows below it, by twelve
// rank-4 steps whose trailing updates are single v_mfma_f64_16x16x4_f64 instructions (K = 4 is the
// MFMA's k extent).  A workgroup (4 waves) keeps 7 row tiles of 16 rows -- tiles 0..2 = the diagonal
// block (factored redundantly by every workgroup), tiles 3..6 = its own rows -- times 3 column tiles
// in MFMA accumulators, transposed: accumulator element (lane, reg) of tile (rt, ct) is matrix
// element (row 16 rt + (lane & 15), column 16 ct + (lane >> 4) + 4 reg), so that global loads and
// stores touch 128 contiguous bytes per 16 lanes.  Per step q (columns 4q..4q+3):
//   A  every lane publishes its one element of the four current columns to LDS;
//   B  every lane factors the 4x4 pivot block redundantly in registers (no cross-lane traffic),
//      lanes 0..111 turn their own row into W = row M4^T (= L D) and L = W D^-1, write the MFMA
//      operands to LDS and stream L21 / W21 / L11 / D to global memory;
//   C  acc(rt, ct) += L_op(ct) x (-W_op(rt))  -- one MFMA per live tile.
// ------------------------------------------------------------------------------------------------
constexpr int PANEL_ROWS = 64;                     // rows below the diagonal block per workgroup
constexpr int PANEL_LR = NB + PANEL_ROWS;          // local rows: 48 diagonal + 64 own = 7 tiles of 16
static_assert(PANEL_LR == 112 && NB == 48, "k_ldl_panel is written for 3 + 4 row tiles of 16");

__device__ __forceinline__ double rcp_nr(double d) {        // 1/d: v_rcp_f64 + two Newton steps (<= 1 ulp)
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return (fabs(d) > DBL_MIN) ? x : 0.0;                      // Eigen's D^+ rule for a vanished pivot
}

__global__ __launch_bounds__(256) void k_ldl_panel(double *__restrict__ A, int nA, int c0, int nR,
                                                   double *__restrict__ dvec, double *__restrict__ Wp,
                                                   double *__restrict__ zvec) {
  __shared__ double cur[4][PANEL_LR];     // the four current columns, all local rows
  __shared__ double Wop[4][PANEL_LR];     // -W (B operand), k-major
  __shared__ double Lop[4][PANEL_LR];     //  L (A operand; only local rows < NB are read)
  const int ldA = 2 * nA + NB;           // nR = live rows: matrix, right-hand side tile, identity tiles <= this panel
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rbase = c0 + NB + blockIdx.x * PANEL_ROWS;        // first own row
  auto grow = [&](int lr) { return lr < NB ? c0 + lr : rbase + (lr - NB); };   // local -> global row

  // ---- load the 7 x 3 tiles into accumulators (wave w owns row tiles w and w + 4) ----
  d4 acc[2][3];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int rt = wv + 4 * s;
    const int gr = grow(rt * 16 + l15);
    const bool ok = rt < 7 && gr < nR;
#pragma unroll
    for (int ct = 0; ct < 3; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        acc[s][ct][e] = ok ? A[(size_t)(c0 + ct * 16 + l4 + 4 * e) * ldA + (ok ? gr : 0)] : 0.0;
  }

#pragma unroll
  for (int q = 0; q < NB / 4; q++) {
    const int jt = q >> 2, qq = q & 3;
    // ---- A: publish columns 4q..4q+3 (lane holds row rt*16 + l15, column 4q + l4) ----
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int rt = wv + 4 * s;
      if (rt < 7) cur[l4][rt * 16 + l15] = acc[s][jt][qq];
    }
    __syncthreads();
    // ---- B: 4x4 pivot block (rows 4q..4q+3 of the diagonal block), redundantly per lane ----
    const int p0 = 4 * q;
    const double a00 = cur[0][p0], a10 = cur[0][p0 + 1], a20 = cur[0][p0 + 2], a30 = cur[0][p0 + 3];
    const double a11 = cur[1][p0 + 1], a21 = cur[1][p0 + 2], a31 = cur[1][p0 + 3];
    const double a22 = cur[2][p0 + 2], a32 = cur[2][p0 + 3], a33 = cur[3][p0 + 3];
    const double d0 = a00, i0 = rcp_nr(d0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = __builtin_fma(-l10, a10, a11), i1 = rcp_nr(d1);
    const double w21 = __builtin_fma(-l20, a10, a21), w31 = __builtin_fma(-l30, a10, a31);
    const double l21 = w21 * i1, l31 = w31 * i1;
    const double d2 = __builtin_fma(-l21, w21, __builtin_fma(-l20, a20, a22)), i2 = rcp_nr(d2);
    const double w32 = __builtin_fma(-l31, w21, __builtin_fma(-l30, a20, a32));
    const double l32 = w32 * i2;
    const double d3 = __builtin_fma(-l32, w32, __builtin_fma(-l31, w31, __builtin_fma(-l30, a30, a33)));
    const double i3 = rcp_nr(d3);
    // M4 = L4^-1 (unit lower)
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = __builtin_fma(l21, l10, -l20);
    const double m31 = __builtin_fma(l32, l21, -l31);
    const double m30 = -l30 - l31 * m10 - l32 * m20;
    if (tid < PANEL_LR) {
      const int lr = tid;
      const double r0 = cur[0][lr], r1 = cur[1][lr], r2 = cur[2][lr], r3 = cur[3][lr];
      double w0 = 0, w1 = 0, w2 = 0, w3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
      const bool below = lr > p0 + 3;                      // rows still to be eliminated
      if (below) {
        w0 = r0;
        w1 = __builtin_fma(m10, r0, r1);
        w2 = __builtin_fma(m21, r1, __builtin_fma(m20, r0, r2));
        w3 = __builtin_fma(m32, r2, __builtin_fma(m31, r1, __builtin_fma(m30, r0, r3)));
        e0 = w0 * i0; e1 = w1 * i1; e2 = w2 * i2; e3 = w3 * i3;
      }
      Wop[0][lr] = -w0; Wop[1][lr] = -w1; Wop[2][lr] = -w2; Wop[3][lr] = -w3;
      Lop[0][lr] = e0; Lop[1][lr] = e1; Lop[2][lr] = e2; Lop[3][lr] = e3;
      // ---- stream the results out ----
      const int gr = grow(lr);
      const size_t cA = (size_t)(c0 + p0) * ldA + gr;
      if (lr >= NB) {                                       // own rows: L21 in place, W21 for ldl_trail
        if (gr < nR) {
          A[cA] = e0; A[cA + ldA] = e1; A[cA + 2 * (size_t)ldA] = e2; A[cA + 3 * (size_t)ldA] = e3;
          const size_t cW = (size_t)p0 * ldA + gr;
          Wp[cW] = w0; Wp[cW + ldA] = w1; Wp[cW + 2 * (size_t)ldA] = w2; Wp[cW + 3 * (size_t)ldA] = w3;
          if (gr == nA) {                                   // right-hand side row: z = D^+ L^-1 P b
            zvec[c0 + p0] = e0; zvec[c0 + p0 + 1] = e1; zvec[c0 + p0 + 2] = e2; zvec[c0 + p0 + 3] = e3;
          }
        }
      } else if (blockIdx.x == 0) {                         // diagonal block: L11 and D, once
        if (below) {
          A[cA] = e0; A[cA + ldA] = e1; A[cA + 2 * (size_t)ldA] = e2; A[cA + 3 * (size_t)ldA] = e3;
        } else if (lr >= p0) {                              // the pivot rows themselves: unit lower L4, D
          const int e = lr - p0;
          if (e >= 1) A[cA] = e == 1 ? l10 : (e == 2 ? l20 : l30);
          if (e >= 2) A[cA + ldA] = e == 2 ? l21 : l31;
          if (e >= 3) A[cA + 2 * (size_t)ldA] = l32;
          dvec[c0 + lr] = e == 0 ? d0 : (e == 1 ? d1 : (e == 2 ? d2 : d3));
        }
      }
    }
    __syncthreads();
    // ---- C: rank-4 update of every live tile: one MFMA each ----
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int rt = wv + 4 * s;
      if (rt < 7) {
        const double bop = Wop[l4][rt * 16 + l15];
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
          if (ct >= jt) {
            const double aop = Lop[l4][ct * 16 + l15];
            acc[s][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[s][ct], 0, 0, 0);
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ldl_trail: A22 -= W21 L21^T on the lower triangle; one wavefront per 48 (rows i) x 16 (cols j)
// tile, f64 MFMA, every operand prefetched before the first MFMA (the kernel is latency-bound).
//   D[m][nn] = sum_k L21[j0+m][k] * W21[i0+nn][k]  -> element (i0+nn, j0+m); lanes of a row group
//   touch 128 contiguous bytes of a column.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ldl_trail(double *__restrict__ A, int nA, int c0,
                                                   const double *__restrict__ Wp, int mt) {
  const int ldA = 2 * nA + NB;
  const int lane = threadIdx.x & 63;
  const int ti = blockIdx.y;                                  // 48-row tile; ti == mt: right-hand side tile
  const int tj = blockIdx.x * 4 + (threadIdx.x >> 6);         // 16-col tile
  if (ti < mt ? (tj > 3 * ti + 2) : (tj >= 3 * mt)) return;
  const int base = c0 + NB;
  const int i0 = base + ti * NB, j0 = base + tj * 16;
  const double *pl = A + (size_t)(c0 + (lane >> 4)) * ldA + j0 + (lane & 15);    // L21 rows j
  const double *pw = Wp + (size_t)(lane >> 4) * ldA + i0 + (lane & 15);          // W21 rows i
  double a[NB / 4], b[NB / 4][3];
#pragma unroll
  for (int ks = 0; ks < NB / 4; ks++) {
    a[ks] = pl[(size_t)ks * 4 * ldA];
#pragma unroll
    for (int q = 0; q < 3; q++) b[ks][q] = pw[(size_t)ks * 4 * ldA + 16 * q];
  }
  // the tile itself (read-modify-write), also in flight before the MFMAs
  double old[3][4];
#pragma unroll
  for (int y = 0; y < 3; y++)
#pragma unroll
    for (int e = 0; e < 4; e++)
      old[y][e] = A[(size_t)(j0 + (lane >> 4) + 4 * e) * ldA + i0 + 16 * y + (lane & 15)];
  d4 acc[3];
#pragma unroll
  for (int y = 0; y < 3; y++) acc[y] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < NB / 4; ks++)
#pragma unroll
    for (int y = 0; y < 3; y++) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks][y], acc[y], 0, 0, 0);
#pragma unroll
  for (int y = 0; y < 3; y++)
#pragma unroll
    for (int e = 0; e < 4; e++)
      A[(size_t)(j0 + (lane >> 4) + 4 * e) * ldA + i0 + 16 * y + (lane & 15)] = old[y][e] - acc[y][e];
}

// ------------------------------------------------------------------------------------------------
// backward solve  L^T x = z  as one product with the factor's own by-product: the identity rows appended to the
// matrix came out of the factorisation as M = L^-T D^+ (row r, columns c >= r), so x = M (D z).
// 64 rows per workgroup, the column range split over its four waves; lanes of a wave read 512 contiguous bytes.
// ------------------------------------------------------------------------------------------------
constexpr int APPLY_CHUNKS = 16;        // column chunks of the product; k_ldl_finish adds the partials in order
__global__ __launch_bounds__(256) void k_ldl_apply(const double *__restrict__ A, int nA, const double *__restrict__ dvec,
                                                   const double *__restrict__ z, double *__restrict__ xpart) {
  __shared__ double sq[256];
  const int ldA = 2 * nA + NB;
  const int rl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, r = r0 + rl;
  // this workgroup: rows r0..r0+63 x column chunk blockIdx.y (multiples of 16 columns; only columns >= r0 matter)
  const int span = ((nA - r0) / 16 + APPLY_CHUNKS - 1) / APPLY_CHUNKS * 16;
  const int cbeg = r0 + blockIdx.y * span, cend = min(nA, cbeg + span);
  double acc0 = 0.0, acc1 = 0.0;
  const double *Mr = A + (size_t)(nA + NB) + min(r, nA - 1);
  for (int cb = cbeg + 16 * q; cb < cend; cb += 64) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const int c = cb + k;
      acc0 = __builtin_fma(Mr[(size_t)c * ldA], dvec[c] * z[c], acc0);
      acc1 = __builtin_fma(Mr[(size_t)(c + 1) * ldA], dvec[c + 1] * z[c + 1], acc1);
    }
  }
  sq[threadIdx.x] = acc0 + acc1;
  __syncthreads();
  if (q == 0 && r < nA) xpart[(size_t)blockIdx.y * nA + r] = (sq[rl] + sq[64 + rl]) + (sq[128 + rl] + sq[192 + rl]);
}

// un-permute, q1 = 0.5 dx.(u D dx - g)    (bavoxel.hpp:1127)
__global__ __launch_bounds__(1024) void k_ldl_finish(const double *__restrict__ x, int nA, int n,
                                                     const int *__restrict__ perm, const double *__restrict__ H,
                                                     const double *__restrict__ g, double u,
                                                     double *__restrict__ dx, double *__restrict__ scal) {
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double q = 0.0;
  for (int r = tid; r < nA; r += 1024) {
    const int p = perm[r];
    if (p < n) {
      double xv = 0.0;
#pragma unroll
      for (int k = 0; k < APPLY_CHUNKS; k++) xv += x[(size_t)k * nA + r];       // partial products of k_ldl_apply
      dx[p] = xv;
      q += xv * (u * H[(size_t)p * n + p] * xv - g[p]);
    }
  }
  red[tid] = q;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) scal[2] = 0.5 * red[0];
}

void launch_solve(balm_ctx *c, double u, bool new_hessian) {
  hipStream_t s = c->stream;
  const int n = c->n, nA = c->nA;
  if (new_hessian)
    hipLaunchKernelGGL(k_rank_diag, dim3((nA + 15) / 16), dim3(256), (size_t)nA * sizeof(double) + 256 * sizeof(int),
                       s, c->d_H, n, nA, c->d_perm);
  {
    long total = (long)(2 * nA + NB) * nA;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_build_A, dim3(grid), dim3(256), 0, s, c->d_H, c->d_g, n, nA, c->d_perm, u, c->d_A);
  }
  const int P = nA / NB;
  for (int p = 0; p < P; p++) {
    const int c0 = p * NB;
    const int m = nA - c0 - NB;                 // square part still to factor
    const int nR = nA + NB + NB * (p + 1);      // live rows: + right-hand side tile + identity tiles 0..p
    const int rows = nR - (c0 + NB);
    hipLaunchKernelGGL(k_ldl_panel, dim3((rows + PANEL_ROWS - 1) / PANEL_ROWS), dim3(256), 0, s, c->d_A, nA, c0, nR,
                       c->d_dvec, c->d_Wp, c->d_z);
    if (m > 0) {
      const int mt = m / NB;
      hipLaunchKernelGGL(k_ldl_trail, dim3((3 * mt + 3) / 4, mt + 1 + (p + 1)), dim3(256), 0, s, c->d_A, nA, c0, c->d_Wp, mt);
    }
  }
  hipLaunchKernelGGL(k_ldl_apply, dim3((nA + 63) / 64, APPLY_CHUNKS), dim3(256), 0, s, c->d_A, nA, c->d_dvec, c->d_z, c->d_x);
  hipLaunchKernelGGL(k_ldl_finish, dim3(1), dim3(1024), 0, s, c->d_x, nA, n, c->d_perm, c->d_H, c->d_g, u, c->d_dx,
                     c->d_scal);
}

// ------------------------------------------------------------------------------------------------
// pose update: left  R <- Exp(dth) R, p <- Exp(dth) p + dt   (bavoxel.hpp:1123-1125)
//              right R <- R Exp(dth), p <- p + dt            (bavoxel.hpp:1119-1120)
// Exp = Rodrigues with the reference's 1e-11 threshold (include/tools.hpp:56-71)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void exp_so3(const double w[3], double E[3][3]) {
  const double nn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  E[0][0] = E[1][1] = E[2][2] = 1.0;
  E[0][1] = E[0][2] = E[1][0] = E[1][2] = E[2][0] = E[2][1] = 0.0;
  if (nn >= 1e-11) {
    const double x = w[0] / nn, y = w[1] / nn, z = w[2] / nn;
    const double K[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
    const double s = sin(nn), c1 = 1.0 - cos(nn);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // evaluation order of the reference's `I33 + sin*K + (1-cos)*K*K`: ((1-cos)*K)*K
        const double kk = (c1 * K[r][0]) * K[0][c] + (c1 * K[r][1]) * K[1][c] + (c1 * K[r][2]) * K[2][c];
        E[r][c] = (E[r][c] + s * K[r][c]) + kk;
      }
  }
}

__global__ void k_update_poses(int form, int W, const double *__restrict__ poses, const double *__restrict__ dx,
                               double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= W) return;
  const double *q = poses + 12 * j;
  const double w[3] = {dx[6 * j], dx[6 * j + 1], dx[6 * j + 2]};
  const double dt[3] = {dx[6 * j + 3], dx[6 * j + 4], dx[6 * j + 5]};
  double E[3][3];
  exp_so3(w, E);
  double R[3][3], p[3] = {q[9], q[10], q[11]};
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R[r][c] = q[3 * c + r];
  double Rn[3][3], pn[3];
  if (form == 0) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) Rn[r][c] = E[r][0] * R[0][c] + E[r][1] * R[1][c] + E[r][2] * R[2][c];
      pn[r] = E[r][0] * p[0] + E[r][1] * p[1] + E[r][2] * p[2] + dt[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) Rn[r][c] = R[r][0] * E[0][c] + R[r][1] * E[1][c] + R[r][2] * E[2][c];
      pn[r] = p[r] + dt[r];
    }
  }
  double *o = out + 12 * j;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o[3 * c + r] = Rn[r][c];
  o[9] = pn[0]; o[10] = pn[1]; o[11] = pn[2];
}

void launch_update_poses(hipStream_t s, int form, int W, const double *poses, const double *dx, double *out) {
  hipLaunchKernelGGL(k_update_poses, dim3((W + 127) / 128), dim3(128), 0, s, form, W, poses, dx, out);
}

// bavoxel.hpp:1159-1164: p_j <- R_0^T (p_j - p_0), R_j <- R_0^T R_j  (pose 0 included)
__global__ __launch_bounds__(256) void k_reanchor(int W, double *__restrict__ poses) {
  __shared__ double e0[12];
  if (threadIdx.x < 12) e0[threadIdx.x] = poses[threadIdx.x];
  __syncthreads();
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    double *q = poses + 12 * j;
    double R[3][3], p[3] = {q[9] - e0[9], q[10] - e0[10], q[11] - e0[11]};
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) R[r][c] = q[3 * c + r];
    // R0^T(r,k) = R0(k,r) = e0[3*r + k]
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        q[3 * c + r] = e0[3 * r] * R[0][c] + e0[3 * r + 1] * R[1][c] + e0[3 * r + 2] * R[2][c];
      q[9 + r] = e0[3 * r] * p[0] + e0[3 * r + 1] * p[1] + e0[3 * r + 2] * p[2];
    }
  }
}

void launch_reanchor(hipStream_t s, int W, double *poses) {
  hipLaunchKernelGGL(k_reanchor, dim3(1), dim3(256), 0, s, W, poses);
}

}  // namespace balm
