// Damped LM solve on gfx950:  (H + u diag H) dx = -g ,  q1 = 0.5 dx.(u D dx - g)
//   reference: src/benchmark/bavoxel.hpp:1113-1114,1127  (Eigen `.ldlt().solve()`)
//
// Eigen's LDLT (in-place lower, unblocked) searches its pivot on the NOT-yet-updated trailing
// diagonal (its update is left-looking), i.e. the elimination order is simply "decreasing
// |diagonal| of the input matrix".  We therefore apply that static symmetric permutation once and
// run an un-pivoted *blocked* LDL^T (D diagonal, possibly negative: the exact second-order Hessian
// is indefinite away from the optimum, SURVEY.md finding 4):
//   per panel of NB=48 columns:  ldl_panel (single-wave workgroups)   L11, D11 in registers (redundantly per
//                                                                     workgroup), W21 = L21 D11, L21
//                                ldl_trail (f64 MFMA, 48x16 tiles)    A22 -= L21 W21^T
//   The right-hand side rides along as one extra row below the matrix (row nA of the ldA = nA+NB
//   row storage): the panel / trail kernels then produce z = D^+ L^-1 P b for free, and only the
//   backward solve  L^T x = z  (+ un-permute, q1) is left for one workgroup in 16-column sub-panels.
// Pose update kernels (bavoxel.hpp:1116-1126, 1159-1164) live here too.
#include <cfloat>

#include "balm_internal.h"

namespace balm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// permutation by decreasing |diag H| (ties by index); padded positions (>= n) go last.
// 64 ranks per workgroup, the j-range split over the four waves.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank_diag(const double *__restrict__ H, int n, int nA,
                                                   int *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) double dabs[];   // [nA] then int part[256]
  int *part = reinterpret_cast<int *>(dabs + nA);
  for (int i = threadIdx.x; i < nA; i += blockDim.x) dabs[i] = i < n ? fabs(H[(size_t)i * n + i]) : -1.0;
  __syncthreads();
  const int il = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  int rank = 0;
  if (i < nA) {
    const double di = dabs[i];
    const int chunk = (nA + 3) / 4;
    const int j0 = q * chunk, j1 = min(nA, j0 + chunk);
    for (int j = j0; j < j1; j++) {
      const double dj = dabs[j];
      rank += (dj > di) || (dj == di && j < i);
    }
  }
  part[threadIdx.x] = rank;
  __syncthreads();
  if (q == 0 && i < nA) {
    rank = part[il] + part[64 + il] + part[128 + il] + part[192 + il];
    if (dabs[i] != dabs[i]) rank = i;   // NaN: keep it somewhere valid; the solve is garbage anyway
    perm[rank] = i;
  }
}

__global__ __launch_bounds__(256) void k_build_A(const double *__restrict__ H, const double *__restrict__ g, int n,
                                                 int nA, const int *__restrict__ perm, double u,
                                                 double *__restrict__ A) {
  const int ldA = nA + NB;
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
    } else {
      v = (r == nA && pc < n) ? -g[pc] : 0.0;     // right-hand side row: P (-JacT)
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
// ldl_panel: one single-wavefront workgroup per 64 rows below the diagonal block.  Every
// workgroup first factors the NB x NB diagonal block redundantly *in registers* (lane j owns the
// full symmetric column j; pivots and multipliers travel by v_readlane, no LDS, no barriers), then
// each lane forward-substitutes its own row:  w L11^T = a  ->  W21 = L21 D11, L21 = W21 D11^-1.
// Workgroup 0 also stores L11 and D11.  (The diagonal block is fully symmetric-valid: build_A
// fills both triangles and ldl_trail updates whole tiles.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ldl_panel(double *__restrict__ A, int nA, int c0,
                                                  double *__restrict__ dvec, double *__restrict__ Wp) {
  __shared__ double Lt[NB * NB];      // L11[j][k] at k*NB + j (j > k): column k contiguous
  const int ldA = nA + NB, nR = nA + NB;          // rows nA..nA+NB-1: right-hand side tile
  const int lane = threadIdx.x;
  const int r = c0 + NB + blockIdx.x * 64 + lane;
  const bool has_row = r < nR;
  const int rr = has_row ? r : nR - 1;
  // this lane's row of A21 (issued first so the loads fly under the factorization)
  double a[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) a[k] = A[(size_t)(c0 + k) * ldA + rr];

  double c[NB];
  const int jc = lane < NB ? lane : 0;          // lanes 48..63 shadow column 0 and never contribute
  const double *colp = A + (size_t)(c0 + jc) * ldA + c0;
#pragma unroll
  for (int i = 0; i < NB; i++) c[i] = colp[i];

#pragma unroll
  for (int k = 0; k < NB; k++) {
    const double dk = bcast(c[k], k);
    const double inv = (fabs(dk) > DBL_MIN) ? 1.0 / dk : 0.0;
    const double f = (lane > k && lane < NB) ? c[k] * inv : 0.0;     // l_jk for the columns still active
#pragma unroll
    for (int i = k + 1; i < NB; i++) {
      const double aik = bcast(c[i], k);                             // a_ik (column k is final up to scale)
      c[i] = __builtin_fma(-aik, f, c[i]);
    }
    __builtin_amdgcn_sched_barrier(0);     // keep hipcc from hoisting hundreds of readlanes
  }
  // lane j: c[j] = d_j, c[i>j] = l_ij d_j
  double dj = 1.0;
#pragma unroll
  for (int i = 0; i < NB; i++) dj = (i == lane) ? c[i] : dj;
  const double invd = (fabs(dj) > DBL_MIN) ? 1.0 / dj : 0.0;
  if (lane < NB) {
#pragma unroll
    for (int i = 1; i < NB; i++) Lt[lane * NB + i] = c[i] * invd;    // rows i <= lane are never read
  }
  if (blockIdx.x == 0 && lane < NB) {
    dvec[c0 + lane] = dj;
    double *colw = A + (size_t)(c0 + lane) * ldA + c0;
#pragma unroll
    for (int i = 1; i < NB; i++)
      if (i > lane) colw[i] = c[i] * invd;
  }
  __syncthreads();
  // forward substitution along the row (right-looking): a_j -= a_k L11[j][k]; L11[j][k] is a
  // wave-uniform LDS broadcast.  hipcc would otherwise hoist ALL 1128 LDS reads above the loop and
  // spill them; the opaque zero tied to an already-final pivot bounds the read lookahead to 2 steps.
#pragma unroll
  for (int k = 0; k < NB - 1; k++) {
    int z = 0;
    asm volatile("" : "+v"(z) : "v"(a[k > 1 ? k - 2 : 0]));
    const double *Lk = Lt + k * NB + z;
#pragma unroll
    for (int j = k + 1; j < NB; j++) a[j] = __builtin_fma(-a[k], Lk[j], a[j]);
  }
#pragma unroll
  for (int j = 0; j < NB; j++) {
    const double dd = bcast(invd, j);       // every lane still active: readlane needs lane j live
    if (has_row) {
      Wp[(size_t)j * ldA + r] = a[j];
      A[(size_t)(c0 + j) * ldA + r] = a[j] * dd;
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
  const int ldA = nA + NB;
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
// backward solve  L^T x = z  (z = row nA of the factored storage) + un-permute + q1; one workgroup
// of 8 waves.  Left-looking over panels of NB columns, descending:
//   t_k = sum_{r below} L[r][c0+k] x[r]   six columns per wave, lanes stride the rows (every load is
//                                        512 contiguous bytes of a column), shuffle reduction;
//   wave 0 then solves the NB x NB unit-triangular block by v_readlane substitution (lane = column).
// ------------------------------------------------------------------------------------------------
constexpr int TPB = 512;
static_assert(NB == 6 * (TPB / 64), "six panel columns per wave");

__global__ __launch_bounds__(TPB) void k_ldl_backsolve(const double *__restrict__ A, int nA, int n,
                                                       const int *__restrict__ perm, const double *__restrict__ H,
                                                       const double *__restrict__ g, double u,
                                                       double *__restrict__ dx, double *__restrict__ scal) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double *y = sh;            // [nA]  z on entry, x on exit
  double *t = sh + nA;       // [NB]
  double *red = t + NB;      // [TPB]
  const int ldA = nA + NB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int P = nA / NB;
  for (int r = tid; r < nA; r += TPB) y[r] = A[(size_t)r * ldA + nA];
  __syncthreads();
  for (int p = P - 1; p >= 0; p--) {
    const int c0 = p * NB;
    // wave 0: column `lane` of the diagonal block, rows below its diagonal (in flight under the dots)
    double Lb[NB];
    if (wv == 0) {
      const double *col = A + (size_t)(c0 + (lane < NB ? lane : 0)) * ldA + c0;
#pragma unroll
      for (int k = 0; k < NB; k++) Lb[k] = col[k];
    }
    double acc[6];
#pragma unroll
    for (int q = 0; q < 6; q++) acc[q] = 0.0;
    const double *cb = A + (size_t)(c0 + 6 * wv) * ldA;
#pragma unroll 4
    for (int r = c0 + NB + lane; r < nA; r += 64) {
      const double xr = y[r];
#pragma unroll
      for (int q = 0; q < 6; q++) acc[q] = __builtin_fma(cb[(size_t)q * ldA + r], xr, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 6; q++) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 6; q++) t[6 * wv + q] = acc[q];
    }
    __syncthreads();
    if (wv == 0) {
      double v = lane < NB ? y[c0 + lane] - t[lane] : 0.0;
#pragma unroll
      for (int k = NB - 1; k >= 1; k--) {
        const double xk = bcast(v, k);
        if (lane < k) v = __builtin_fma(-Lb[k], xk, v);
      }
      if (lane < NB) y[c0 + lane] = v;
    }
    __syncthreads();
  }
  // un-permute, q1 = 0.5 dx.(u D dx - g)    (bavoxel.hpp:1127)
  double q = 0.0;
  for (int r = tid; r < nA; r += TPB) {
    const int p = perm[r];
    if (p < n) {
      const double x = y[r];
      dx[p] = x;
      q += x * (u * H[(size_t)p * n + p] * x - g[p]);
    }
  }
  red[tid] = q;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) scal[2] = 0.5 * red[0];
}

void launch_solve(balm_ctx *c, double u, bool new_hessian) {
  hipStream_t s = c->stream;
  const int n = c->n, nA = c->nA;
  if (new_hessian)
    hipLaunchKernelGGL(k_rank_diag, dim3((nA + 63) / 64), dim3(256), (size_t)nA * sizeof(double) + 256 * sizeof(int),
                       s, c->d_H, n, nA, c->d_perm);
  {
    long total = (long)(nA + NB) * nA;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_build_A, dim3(grid), dim3(256), 0, s, c->d_H, c->d_g, n, nA, c->d_perm, u, c->d_A);
  }
  const int P = nA / NB;
  for (int p = 0; p < P; p++) {
    const int c0 = p * NB;
    const int m = nA - c0 - NB;                 // square part still to factor
    const int rows = m + NB;                    // + right-hand side tile
    hipLaunchKernelGGL(k_ldl_panel, dim3((rows + 63) / 64), dim3(64), 0, s, c->d_A, nA, c0, c->d_dvec, c->d_Wp);
    if (m > 0) {
      const int mt = m / NB;
      hipLaunchKernelGGL(k_ldl_trail, dim3((3 * mt + 3) / 4, mt + 1), dim3(256), 0, s, c->d_A, nA, c0, c->d_Wp, mt);
    }
  }
  size_t lds = (size_t)(nA + NB + TPB) * sizeof(double);
  hipLaunchKernelGGL(k_ldl_backsolve, dim3(1), dim3(TPB), lds, s, c->d_A, nA, n, c->d_perm, c->d_H, c->d_g, u, c->d_dx,
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
