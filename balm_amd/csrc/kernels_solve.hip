// Damped LM solve on gfx950:  (H + u diag H) dx = -g ,  q1 = 0.5 dx.(u D dx - g)
//   reference: src/benchmark/bavoxel.hpp:1113-1114,1127  (Eigen `.ldlt().solve()`)
//
// Eigen's LDLT (in-place lower, unblocked) searches its pivot on the NOT-yet-updated trailing
// diagonal (its update is left-looking), i.e. the elimination order is simply "decreasing
// |diagonal| of the input matrix".  We therefore apply that static symmetric permutation once and
// run an un-pivoted *blocked* LDL^T (D diagonal, possibly negative: the exact second-order Hessian
// is indefinite away from the optimum, SURVEY.md finding 4):
//   per panel of NB=48 columns:  ldl_diag  (one workgroup, LDS)       L11, D11, M = L11^-1
//                                ldl_panel (row blocks)               W21 = A21 M^T, L21 = W21 D11^-1
//                                ldl_trail (f64 MFMA, 48x48 tiles)    A22 -= L21 W21^T
//   then one workgroup does  P b -> L^-1 -> D^+ -> L^-T -> P^T  using the stored M blocks.
// Pose update kernels (bavoxel.hpp:1116-1126, 1159-1164) live here too.
#include <cfloat>

#include "balm_internal.h"

namespace balm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// permutation by decreasing |diag H| (ties by index); padded positions (>= n) go last
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rank_diag(const double *__restrict__ H, int n, int nA,
                                                    int *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) double dabs[];
  for (int i = threadIdx.x; i < nA; i += blockDim.x) dabs[i] = i < n ? fabs(H[(size_t)i * n + i]) : -1.0;
  __syncthreads();
  for (int i = threadIdx.x; i < nA; i += blockDim.x) {
    const double di = dabs[i];
    int rank = 0;
    for (int j = 0; j < nA; j++) {
      const double dj = dabs[j];
      rank += (dj > di) || (dj == di && j < i);
    }
    if (di != di) rank = i;   // NaN: keep it somewhere valid; the solve is garbage anyway
    perm[rank] = i;
  }
}

__global__ __launch_bounds__(256) void k_build_A(const double *__restrict__ H, int n, int nA,
                                                 const int *__restrict__ perm, double u, double *__restrict__ A) {
  const long total = (long)nA * nA;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / nA), r = (int)(t - (long)c * nA);
    const int pr = perm[r], pc = perm[c];
    double v;
    if (pr < n && pc < n) {
      v = H[(size_t)pc * n + pr];
      if (r == c) v += u * v;      // D = diag(H)  (bavoxel.hpp:1113)
    } else {
      v = (r == c) ? 1.0 : 0.0;
    }
    A[t] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// ldl_diag: unblocked LDL^T of the NB x NB diagonal block in LDS + inverse of its unit-lower factor
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ldl_diag(double *__restrict__ A, int nA, int c0, double *__restrict__ dvec,
                                                  double *__restrict__ Minv) {
  __shared__ double S[NB][NB + 1];
  __shared__ double Mi[NB][NB + 1];
  __shared__ double lcol[NB];
  const int tid = threadIdx.x;
  for (int t = tid; t < NB * NB; t += 256) {
    const int c = t / NB, r = t - c * NB;
    S[r][c] = (r >= c) ? A[(size_t)(c0 + c) * nA + c0 + r] : 0.0;
  }
  __syncthreads();
  for (int k = 0; k < NB; k++) {
    const double d = S[k][k];
    const double inv = (fabs(d) > DBL_MIN) ? 1.0 / d : 0.0;
    if (tid > k && tid < NB) lcol[tid] = S[tid][k] * inv;
    __syncthreads();
    // trailing update of the lower triangle: S[i][j] -= l_i * a_jk  (k < j <= i)
    const int m = NB - 1 - k;
    for (int t = tid; t < m * m; t += 256) {
      const int ii = t / m, jj = t - ii * m;
      if (jj <= ii) {
        const int i = k + 1 + ii, j = k + 1 + jj;
        S[i][j] -= lcol[i] * S[j][k];
      }
    }
    __syncthreads();
    if (tid > k && tid < NB) S[tid][k] = lcol[tid];
    // (next iteration reads S[k+1][k+1] and column k+1 only; column k is final -> no hazard
    //  with the write above because the sync at the top of the next update phase orders it)
  }
  __syncthreads();
  // M = L11^-1 (unit lower), column j by lane j
  if (tid < NB) {
    const int j = tid;
    for (int r = 0; r < NB; r++) Mi[r][j] = (r == j) ? 1.0 : 0.0;
    for (int r = j + 1; r < NB; r++) {
      double s = 0.0;
      for (int k = j; k < r; k++) s += S[r][k] * Mi[k][j];
      Mi[r][j] = -s;
    }
    dvec[c0 + j] = S[j][j];
  }
  __syncthreads();
  double *Mo = Minv + (size_t)(c0 / NB) * NB * NB;
  for (int t = tid; t < NB * NB; t += 256) {
    const int c = t / NB, r = t - c * NB;
    if (r > c) A[(size_t)(c0 + c) * nA + c0 + r] = S[r][c];
    Mo[r * NB + c] = Mi[r][c];     // row-major M
  }
}

// ------------------------------------------------------------------------------------------------
// ldl_panel: W21 = A21 M^T (= L21 D11), L21 = W21 D11^-1 for 64 rows per workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ldl_panel(double *__restrict__ A, int nA, int c0,
                                                   const double *__restrict__ dvec, const double *__restrict__ Minv,
                                                   double *__restrict__ Wp) {
  __shared__ double Ab[64][NB + 1];
  __shared__ double M[NB][NB + 1];
  __shared__ double dinv[NB];
  const int tid = threadIdx.x;
  const int r0 = c0 + NB + blockIdx.x * 64;
  const double *Mo = Minv + (size_t)(c0 / NB) * NB * NB;
  for (int t = tid; t < NB * NB; t += 256) M[t / NB][t % NB] = Mo[t];
  if (tid < NB) {
    const double d = dvec[c0 + tid];
    dinv[tid] = (fabs(d) > DBL_MIN) ? 1.0 / d : 0.0;
  }
  for (int t = tid; t < 64 * NB; t += 256) {
    const int k = t / 64, r = t - k * 64;
    Ab[r][k] = (r0 + r < nA) ? A[(size_t)(c0 + k) * nA + r0 + r] : 0.0;
  }
  __syncthreads();
  const int r = tid & 63, jg = tid >> 6;
  if (r0 + r < nA) {
    for (int j = jg; j < NB; j += 4) {
      double s = 0.0;
      for (int k = 0; k <= j; k++) s += Ab[r][k] * M[j][k];
      Wp[(size_t)j * nA + r0 + r] = s;
      A[(size_t)(c0 + j) * nA + r0 + r] = s * dinv[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ldl_trail: A22 -= W21 L21^T on the lower triangle, 48x48 tile per wavefront, f64 MFMA.
//   D[m][nn] = sum_k L21[j0+m][k] * W21[i0+nn][k]  -> element (i0+nn, j0+m), stored column-major so
//   the 16 lanes of a row group touch 128 contiguous bytes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ldl_trail(double *__restrict__ A, int nA, int c0,
                                                   const double *__restrict__ Wp, int mt, int ntile) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntile) return;
  // t -> (ti >= tj) in the lower triangle of an mt x mt tile grid
  int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  while (ti * (ti + 1) / 2 > t) ti--;
  const int tj = t - ti * (ti + 1) / 2;
  const int base = c0 + NB;
  const int i0 = base + ti * NB, j0 = base + tj * NB;
  d4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const double *pl = A + (size_t)(c0 + (lane >> 4)) * nA + j0 + (lane & 15);    // L21 rows j
  const double *pw = Wp + (size_t)(lane >> 4) * nA + i0 + (lane & 15);          // W21 rows i
#pragma unroll
  for (int ks = 0; ks < NB / 4; ks++) {
    double a[3], b[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      a[q] = pl[(size_t)ks * 4 * nA + 16 * q];
      b[q] = pw[(size_t)ks * 4 * nA + 16 * q];
    }
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
      for (int y = 0; y < 3; y++) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
  }
#pragma unroll
  for (int x = 0; x < 3; x++)
#pragma unroll
    for (int y = 0; y < 3; y++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int j = j0 + 16 * x + (lane >> 4) + 4 * e;
        const int i = i0 + 16 * y + (lane & 15);
        A[(size_t)j * nA + i] -= acc[x][y][e];
      }
}

// ------------------------------------------------------------------------------------------------
// triangular solves + un-permute + q1, one workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ldl_solve(const double *__restrict__ A, int nA, int n,
                                                    const double *__restrict__ dvec,
                                                    const double *__restrict__ Minv, const int *__restrict__ perm,
                                                    const double *__restrict__ H, const double *__restrict__ g,
                                                    double u, double *__restrict__ dx, double *__restrict__ scal) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double *y = sh;            // [nA]
  double *tb = sh + nA;      // [NB]
  double *red = tb + NB;     // [1024]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int P = nA / NB;
  for (int r = tid; r < nA; r += 1024) {
    const int p = perm[r];
    y[r] = p < n ? -g[p] : 0.0;
  }
  __syncthreads();
  // forward: L y' = y
  for (int p = 0; p < P; p++) {
    const int c0 = p * NB;
    const double *M = Minv + (size_t)p * NB * NB;
    if (tid < NB) {
      double s = 0.0;
      for (int k = 0; k <= tid; k++) s += M[tid * NB + k] * y[c0 + k];
      tb[tid] = s;
    }
    __syncthreads();
    if (tid < NB) y[c0 + tid] = tb[tid];
    for (int r = c0 + NB + tid; r < nA; r += 1024) {
      double s = 0.0;
#pragma unroll 8
      for (int k = 0; k < NB; k++) s += A[(size_t)(c0 + k) * nA + r] * tb[k];
      y[r] -= s;
    }
    __syncthreads();
  }
  // D^+   (Eigen LDLT::_solve_impl: zero where |d| <= min())
  for (int r = tid; r < nA; r += 1024) {
    const double d = dvec[r];
    y[r] = (fabs(d) > DBL_MIN) ? y[r] / d : 0.0;
  }
  __syncthreads();
  // backward: L^T x = z
  for (int p = P - 1; p >= 0; p--) {
    const int c0 = p * NB;
    const double *M = Minv + (size_t)p * NB * NB;
    for (int k = wv; k < NB; k += 16) {
      double s = 0.0;
      const double *col = A + (size_t)(c0 + k) * nA;
      for (int r = c0 + NB + lane; r < nA; r += 64) s += col[r] * y[r];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) tb[k] = y[c0 + k] - s;
    }
    __syncthreads();
    if (tid < NB) {
      double s = 0.0;
      for (int k = tid; k < NB; k++) s += M[k * NB + tid] * tb[k];
      y[c0 + tid] = s;
    }
    __syncthreads();
  }
  // un-permute, q1 = 0.5 dx.(u D dx - g)    (bavoxel.hpp:1127)
  double q = 0.0;
  for (int r = tid; r < nA; r += 1024) {
    const int p = perm[r];
    if (p < n) {
      const double x = y[r];
      dx[p] = x;
      q += x * (u * H[(size_t)p * n + p] * x - g[p]);
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
    hipLaunchKernelGGL(k_rank_diag, dim3(1), dim3(1024), (size_t)nA * sizeof(double), s, c->d_H, n, nA, c->d_perm);
  {
    long total = (long)nA * nA;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_build_A, dim3(grid), dim3(256), 0, s, c->d_H, n, nA, c->d_perm, u, c->d_A);
  }
  const int P = nA / NB;
  for (int p = 0; p < P; p++) {
    const int c0 = p * NB;
    hipLaunchKernelGGL(k_ldl_diag, dim3(1), dim3(256), 0, s, c->d_A, nA, c0, c->d_dvec, c->d_Minv);
    const int m = nA - c0 - NB;
    if (m > 0) {
      hipLaunchKernelGGL(k_ldl_panel, dim3((m + 63) / 64), dim3(256), 0, s, c->d_A, nA, c0, c->d_dvec, c->d_Minv,
                         c->d_Wp);
      const int mt = m / NB, ntile = mt * (mt + 1) / 2;
      hipLaunchKernelGGL(k_ldl_trail, dim3((ntile + 3) / 4), dim3(256), 0, s, c->d_A, nA, c0, c->d_Wp, mt, ntile);
    }
  }
  size_t lds = (size_t)(nA + NB + 1024) * sizeof(double);
  hipLaunchKernelGGL(k_ldl_solve, dim3(1), dim3(1024), lds, s, c->d_A, nA, n, c->d_dvec, c->d_Minv, c->d_perm,
                     c->d_H, c->d_g, u, c->d_dx, c->d_scal);
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
        const double kk = K[r][0] * K[0][c] + K[r][1] * K[1][c] + K[r][2] * K[2][c];
        E[r][c] += s * K[r][c] + c1 * kk;
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
