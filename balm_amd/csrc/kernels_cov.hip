// Point-noise -> pose covariance on gfx950 ("next" row N4 of SURVEY.md 8f): the covariance tail of the
// consistency experiment,
//     Rcov_raw = sum_a sum_j Ls_{a,j} c_cov_{a,j} Ls_{a,j}^T     src/simulation/BAs_left.hpp:342-473 (left_jacobian_point)
//     Rcov     = H^-1 Rcov_raw H^-T                              src/simulation/BAs_left.hpp:1089-1096
// The reference builds a dense (6W x 9) Ls per (feature a, observing pose j) and adds a rank-9 6W x 6W update for
// each: O(S W^2) with S = F W observations.  Ls has the structure (derivation: DESIGN.md 7d)
//     block p of Ls_{a,j} = At_{a,p} Gm_{a,j} + [p == j] D_{a,j},   At (6x3), Gm (3x9), D (6x9),
// where the columns of At are the Hessian's own factor vectors rescaled.  Summing over j per feature,
//     Rcov_raw = X X^T - Y Y^T + blockdiag_j(S_j),   X = At Cq + Y,  Y = Rr Cq^-T,  Q = Cq Cq^T,
//     Q = sum_j Gm c_cov Gm^T (3x3),  Rr block j = D c_cov Gm^T (6x3),  S_j = sum_a D c_cov D^T (6x6),
// with X, Y in R^{6W x 3F}: two more launches of the Hessian's FP64 MFMA SYRK (k_hessian_syrk), O(F W^2).
//   k_cov_factors   one workgroup per feature, one lane per pose: X and Y columns, S partials
//   k_hessian_syrk  (kernels_accum.hip) on X, then on Y
//   k_cov_assemble  Rcov_raw = XX^T - YY^T + blockdiag(S)
//   k_tri_gemm      H^-1 Rraw H^-T = M D (M^T Rraw M) D M^T with the factorisation's by-product M = L^-T D^+
#include <cfloat>

#include "balm_internal.h"

namespace balm {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// the three top rows of g1(w) (BAs_left.hpp:320-330): d(Co)[:3,:] w for the 9 noise coordinates
// [Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz]; the fourth row is [0 0 0 0 0 0 w0 w1 w2]
__device__ __forceinline__ void g1_top(const double w[4], double g[3][9]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 9; c++) g[r][c] = 0.0;
  g[0][0] = w[0]; g[0][1] = w[1]; g[0][2] = w[2]; g[0][6] = w[3];
  g[1][1] = w[0]; g[1][3] = w[1]; g[1][4] = w[2]; g[1][7] = w[3];
  g[2][2] = w[0]; g[2][4] = w[1]; g[2][5] = w[2]; g[2][8] = w[3];
}

// y = c_cov x for the c_cov that PointCluster::push accumulates with p_cov = sigma^2 I (toolss.hpp:321-345):
// sum_k Bf Bf^T is linear in the cluster's own moments (P, v, N) and has 45 non-zeros -- no 9x9 matrix is formed
__device__ __forceinline__ void iso_cov_mv(const double P[6], const double v[3], double N, double s2, const double x[9],
                                           double y[9]) {
  const double xx = P[0], xy = P[1], xz = P[2], yy = P[3], yz = P[4], zz = P[5], vx = v[0], vy = v[1], vz = v[2];
  y[0] = s2 * (4 * xx * x[0] + 2 * xy * x[1] + 2 * xz * x[2] + 2 * vx * x[6]);
  y[1] = s2 * (2 * xy * x[0] + (yy + xx) * x[1] + yz * x[2] + 2 * xy * x[3] + xz * x[4] + vy * x[6] + vx * x[7]);
  y[2] = s2 * (2 * xz * x[0] + yz * x[1] + (zz + xx) * x[2] + xy * x[4] + 2 * xz * x[5] + vz * x[6] + vx * x[8]);
  y[3] = s2 * (2 * xy * x[1] + 4 * yy * x[3] + 2 * yz * x[4] + 2 * vy * x[7]);
  y[4] = s2 * (xz * x[1] + xy * x[2] + 2 * yz * x[3] + (zz + yy) * x[4] + 2 * yz * x[5] + vz * x[7] + vy * x[8]);
  y[5] = s2 * (2 * xz * x[2] + 2 * yz * x[4] + 4 * zz * x[5] + 2 * vz * x[8]);
  y[6] = s2 * (2 * vx * x[0] + vy * x[1] + vz * x[2] + N * x[6]);
  y[7] = s2 * (vx * x[1] + 2 * vy * x[3] + vz * x[4] + N * x[7]);
  y[8] = s2 * (vx * x[2] + vy * x[4] + 2 * vz * x[5] + N * x[8]);
}

constexpr int COV_DACC = 21;      // upper triangle of the 6x6 block S_j, row-major

// ------------------------------------------------------------------------------------------------
// One workgroup per feature at a time, one lane per pose (like k_feature_factors).  Phase 1 computes the
// pose's At / Rr rows (parked in the X / Y columns), its share of Q and of S_j; after the block-wide sum of Q
// every lane factors the 3x3 Q redundantly and turns its own rows into X and Y.
// ONEPASS (round 4; windows of up to 256 poses: one pose per lane): the pose's At / Rr rows stay in REGISTERS across the block-wide sum
// instead of being parked in the X / Y columns (written as 8-byte pieces 48 bytes apart, read back the same way, written again), and the
// finished X / Y columns leave through a 3 KB LDS staging block per wavefront as fully coalesced 16-byte stores, as k_feature_factors'
// Gt columns do (kernels_accum.hip).  Same arithmetic, same order: bit for bit the two-pass result.
// ------------------------------------------------------------------------------------------------
// What bounds it (round 5, profiles/r05c_cov_*.txt): registers.  ~150 doubles are alive per observation (At, Rr, Y, c_cov Gm^T, the cluster,
// the pose): 256 VGPRs + 40-118 spilled to AGPRs = ONE wave per SIMD, 1.96 ms at W = 200 / F = 50 000 for 3.7 GB of traffic (0.30 of the
// copy rate; the kernel is latency-bound at that occupancy, not bandwidth- or FP64-bound: ~2 000 flops per observation are 0.3 ms of
// the FP64 pipes).  Two waves per SIMD (at most 256 registers each, 156-530 bytes of scratch per lane instead of the AGPR spills) + At
// computed last: 1.63 ms, bit-identical.  Three are not possible: the [12 + 21][W] doubles of LDS per workgroup cap a CU at two workgroups.
template <bool EXPLICIT, bool ONEPASS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cov_factors(const double *__restrict__ cl, const double *__restrict__ ccov,
                                                     double sigma2, const double *__restrict__ poses,
                                                     const double *__restrict__ feat, int W, int npad, int F,
                                                     double *__restrict__ Gx, double *__restrict__ Gy,
                                                     double *__restrict__ dpart) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *sp = sm;                 // [12][W] poses
  double *sacc = sm + 12 * W;      // [21][W]  (in registers instead -- 42 of them per lane, measured round 6 -- the kernel spills more than it saves:
                                   // the register file, not the LDS, is what holds a CU at two workgroups)
  __shared__ double sq[4][6];
  for (int t = threadIdx.x; t < 12 * W; t += blockDim.x) {
    int i = t / 12, c = t - 12 * i;
    sp[c * W + i] = poses[t];
  }
  for (int t = threadIdx.x; t < COV_DACC * W; t += blockDim.x) sacc[t] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;

  for (int a = blockIdx.x; a < F; a += gridDim.x) {
    const double *f = feat + (size_t)a * FEAT_STRIDE;
    const double NN = f[FT_NN], iNN = 1.0 / NN, coe = f[FT_COE];
    const double vbar[3] = {f[FT_VBAR], f[FT_VBAR + 1], f[FT_VBAR + 2]};
    const double lam[3] = {f[FT_LAM], f[FT_LAM + 1], f[FT_LAM + 2]};
    const double u0[3] = {f[FT_U0], f[FT_U0 + 1], f[FT_U0 + 2]};
    const double u1[3] = {f[FT_U1], f[FT_U1 + 1], f[FT_U1 + 2]};
    const double u2[3] = {f[FT_U2], f[FT_U2 + 1], f[FT_U2 + 2]};
    const double k1 = 1.0 / ((lam[0] - lam[1]) * NN), k2 = 1.0 / ((lam[0] - lam[2]) * NN);
    const double vu0 = vbar[0] * u0[0] + vbar[1] * u0[1] + vbar[2] * u0[2];
    const double *ca = cl + (size_t)a * 10 * W;
    double *gx = Gx + (size_t)(3 * a) * npad, *gy = Gy + (size_t)(3 * a) * npad;
    double q[6] = {0, 0, 0, 0, 0, 0};

    double at[6][3], rr[6][3];               // (ONEPASS: alive until the columns are written)
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) at[r][c] = rr[r][c] = 0.0;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
      if (!ONEPASS) {
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) at[r][c] = rr[r][c] = 0.0;
      }
      const double N = ca[(size_t)9 * W + i];
      if ((int)N > 0) {
        double P[6], v[3], R[9], p[3];
#pragma unroll
        for (int c = 0; c < 6; c++) P[c] = ca[(size_t)c * W + i];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = ca[(size_t)(6 + c) * W + i];
#pragma unroll
        for (int c = 0; c < 9; c++) R[c] = sp[c * W + i];       // column-major: R(r,c) = R[3c+r]
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] = sp[(9 + c) * W + i];
        // world moments of the observation (tools.hpp:333-339)
        double Rv[3], b[3], Pw[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
          Rv[r] = R[r] * v[0] + R[3 + r] * v[1] + R[6 + r] * v[2];
          b[r] = Rv[r] + N * p[r];
        }
        {
          const double Pf[3][3] = {{P[0], P[1], P[2]}, {P[1], P[3], P[4]}, {P[2], P[4], P[5]}};
          double RP[3][3];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) RP[r][c] = R[r] * Pf[0][c] + R[3 + r] * Pf[1][c] + R[6 + r] * Pf[2][c];
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
              Pw[r][c] = RP[r][0] * R[c] + RP[r][1] * R[3 + c] + RP[r][2] * R[6 + c] + Rv[r] * p[c] + p[r] * b[c];
        }
        const double c2 = 2.0 * iNN;
        // Round 6: the sparse structure is used instead of carried.  g1(w) (BAs_left.hpp:320-330) has fifteen non-zeros that four scalars
        // define -- row j of its top: w0, w1, w2 at the coordinates of P's row j, w3 at v_j; bottom row [0 .. 0 w0 w1 w2] -- so
        //     Gm's rows      k_k (ru_k^T g1([r3; p.u0])[:3] + [0 | pv_k r3 - (vbar.u0) ru_k]),  m = [0 .. 0 r3]        (:431-441)
        //     D = (2/NN) U_0 Y,  Y = T_j G1,  G1 = g1([r3; (p - vbar).u0])  (4 x 9),  T_j = [R p; 0 1]                    (:449-450)
        // and everything that round 5 computed on the 3 x 9 Y (27 registers, beside the 27 of c_cov Gm^T) is computed on G1 and lifted
        // through T_j afterwards:   Y sg = T_j (G1 sg),   Y c_cov Y^T = T_j (G1 c_cov G1^T) T_j^T.  One 9-vector is alive at a time.
        double r3[3], ru1[3], ru2[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          r3[c] = R[3 * c] * u0[0] + R[3 * c + 1] * u0[1] + R[3 * c + 2] * u0[2];    // R^T u0
          ru1[c] = R[3 * c] * u1[0] + R[3 * c + 1] * u1[1] + R[3 * c + 2] * u1[2];   // R^T u_k  (u_k^T R)
          ru2[c] = R[3 * c] * u2[0] + R[3 * c + 1] * u2[1] + R[3 * c + 2] * u2[2];
        }
        const double pu = p[0] * u0[0] + p[1] * u0[1] + p[2] * u0[2];
        const double st = pu - vu0;
        // row j of g1_top(w) as a dense 9-vector (the zeros are literals: they cost no register)
        auto g1row = [&](int j, double w3, double x[9]) {
#pragma unroll
          for (int c = 0; c < 9; c++) x[c] = 0.0;
          if (j == 0) { x[0] = r3[0]; x[1] = r3[1]; x[2] = r3[2]; x[6] = w3; }
          else if (j == 1) { x[1] = r3[0]; x[3] = r3[1]; x[4] = r3[2]; x[7] = w3; }
          else { x[2] = r3[0]; x[4] = r3[1]; x[5] = r3[2]; x[8] = w3; }
        };
        // x . (row j of g1_top(w)): four terms
        auto g1dot = [&](int j, double w3, const double x[9]) {
          return j == 0 ? r3[0] * x[0] + r3[1] * x[1] + r3[2] * x[2] + w3 * x[6]
               : j == 1 ? r3[0] * x[1] + r3[1] * x[3] + r3[2] * x[4] + w3 * x[7]
                        : r3[0] * x[2] + r3[1] * x[4] + r3[2] * x[5] + w3 * x[8];
        };
        // products with the cluster's 9x9 noise covariance (symmetric): explicit matrix, or the isotropic closed form
        double cc[EXPLICIT ? 9 : 1][9];
        if (EXPLICIT) {
          const double *src = ccov + ((size_t)a * W + i) * 81;
#pragma unroll
          for (int r = 0; r < 9; r++)
#pragma unroll
            for (int k = 0; k < 9; k++) cc[EXPLICIT ? r : 0][k] = src[9 * r + k];
        }
        auto covmv = [&](const double x[9], double y[9]) {
          if (EXPLICIT) {
#pragma unroll
            for (int r = 0; r < 9; r++) {
              double t = 0.0;
#pragma unroll
              for (int c = 0; c < 9; c++) t += cc[EXPLICIT ? r : 0][c] * x[c];
              y[r] = t;
            }
          } else {
            iso_cov_mv(P, v, N, sigma2, x, y);
          }
        };
        // c_cov Gm^T ONE COLUMN AT A TIME: a column feeds its part of Q (Gm's rows r <= k against it), its column of Rr (through G1 and T_j)
        // and -- the third -- N4's corner, then it is dead: 9 registers instead of round 5's 27.
        // Gm's dense rows: gm_k[c] = k_k (ru_k . g1_top([r3; pu])[:, c] (+ the v-part's extras)); row 2 = [0 .. 0 r3]
        double n33;
        {
          const double pv1 = (p[0] - vbar[0]) * u1[0] + (p[1] - vbar[1]) * u1[1] + (p[2] - vbar[2]) * u1[2];
          const double pv2 = (p[0] - vbar[0]) * u2[0] + (p[1] - vbar[1]) * u2[1] + (p[2] - vbar[2]) * u2[2];
          double gmq[2][9];
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const double *ru = k == 0 ? ru1 : ru2;
            const double kk = k == 0 ? k1 : k2, pv = k == 0 ? pv1 : pv2;
            double *g = gmq[k];
            g[0] = kk * (ru[0] * r3[0]);
            g[1] = kk * (ru[0] * r3[1] + ru[1] * r3[0]);
            g[2] = kk * (ru[0] * r3[2] + ru[2] * r3[0]);
            g[3] = kk * (ru[1] * r3[1]);
            g[4] = kk * (ru[1] * r3[2] + ru[2] * r3[1]);
            g[5] = kk * (ru[2] * r3[2]);
#pragma unroll
            for (int c = 0; c < 3; c++) g[6 + c] = kk * (ru[c] * pu + pv * r3[c] - vu0 * ru[c]);
          }
          const double m9[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, r3[0], r3[1], r3[2]};
          // q's order: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
          const int qi[3][3] = {{0, 1, 2}, {-1, 3, 4}, {-1, -1, 5}};
#pragma unroll
          for (int k = 0; k < 3; k++) {
            double col[9];
            covmv(k == 0 ? gmq[0] : (k == 1 ? gmq[1] : m9), col);
#pragma unroll
            for (int r = 0; r <= k; r++) {
              double sum = 0.0;
              if (r < 2) {
#pragma unroll
                for (int c = 0; c < 9; c++) sum += gmq[r][c] * col[c];
              } else {
                sum = r3[0] * col[6] + r3[1] * col[7] + r3[2] * col[8];
              }
              q[qi[r][k]] += sum;
            }
            // Rr[:, k] = c2 U_0 (T_j (G1 col)):  rows 0..2 = (R (G1top col) + p (r3 . col[6..8])) x u0, rows 3..5 = u0 (r3 . col[6..8])
            const double t0 = g1dot(0, st, col), t1 = g1dot(1, st, col), t2 = g1dot(2, st, col);
            const double y3 = r3[0] * col[6] + r3[1] * col[7] + r3[2] * col[8];
            double ys[3];
#pragma unroll
            for (int r = 0; r < 3; r++) ys[r] = R[r] * t0 + R[3 + r] * t1 + R[6 + r] * t2 + p[r] * y3;
            double yx[3];
            cross3(ys, u0, yx);                  // hat(-u0) y = y x u0
#pragma unroll
            for (int r = 0; r < 3; r++) { rr[r][k] = c2 * yx[r]; rr[3 + r][k] = c2 * u0[r] * y3; }
            if (k == 2) n33 = y3;                // m . c_cov m
          }
        }
        // S_j = D c_cov D^T = c2^2 U_0 M U_0^T,  M = Y4 c_cov Y4^T = T_j N4 T_j^T,  N4 = G1 c_cov G1^T (4x4 symmetric)
        {
          double n00, n01, n02, n03, n11, n12, n13, n22, n23;
          {
            double x[9], f[9];
            g1row(0, st, x); covmv(x, f);
            n00 = g1dot(0, st, f); n01 = g1dot(1, st, f); n02 = g1dot(2, st, f); n03 = r3[0] * f[6] + r3[1] * f[7] + r3[2] * f[8];
            g1row(1, st, x); covmv(x, f);
            n11 = g1dot(1, st, f); n12 = g1dot(2, st, f); n13 = r3[0] * f[6] + r3[1] * f[7] + r3[2] * f[8];
            g1row(2, st, x); covmv(x, f);
            n22 = g1dot(2, st, f); n23 = r3[0] * f[6] + r3[1] * f[7] + r3[2] * f[8];
          }
          const double N3[3][3] = {{n00, n01, n02}, {n01, n11, n12}, {n02, n12, n22}}, n3[3] = {n03, n13, n23};
          double M[4][4];
          {
            double A3[3][3], Rn[3];              // A3 = R N3,  Rn = R n3
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
              for (int c = 0; c < 3; c++) A3[r][c] = R[r] * N3[0][c] + R[3 + r] * N3[1][c] + R[6 + r] * N3[2][c];
              Rn[r] = R[r] * n3[0] + R[3 + r] * n3[1] + R[6 + r] * n3[2];
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
              for (int c = r; c < 3; c++) {
                const double m = A3[r][0] * R[c] + A3[r][1] * R[3 + c] + A3[r][2] * R[6 + c] + Rn[r] * p[c] + p[r] * Rn[c] + n33 * p[r] * p[c];
                M[r][c] = M[c][r] = m;
              }
              M[r][3] = M[3][r] = Rn[r] + n33 * p[r];
            }
            M[3][3] = n33;
          }
          // top-left: K M33 K^T with K = hat(-u0) (K y = y x u0); top-right: (K M[:3][3]) u0^T; bottom-right: M33' = M[3][3] u0 u0^T
          double KM[3][3];                     // column c of K M33: M33[:, c] x u0
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double col[3] = {M[0][c], M[1][c], M[2][c]};
            double x[3];
            cross3(col, u0, x);
            KM[0][c] = x[0]; KM[1][c] = x[1]; KM[2][c] = x[2];
          }
          double S6[6][6];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const double row[3] = {KM[r][0], KM[r][1], KM[r][2]};     // (K M33)[r, :] -> times K^T: row x u0
            double x[3];
            cross3(row, u0, x);
#pragma unroll
            for (int c = 0; c < 3; c++) S6[r][c] = x[c];
          }
          {
            const double m3[3] = {M[0][3], M[1][3], M[2][3]};
            double km3[3];
            cross3(m3, u0, km3);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int c = 0; c < 3; c++) { S6[r][3 + c] = km3[r] * u0[c]; S6[3 + r][3 + c] = M[3][3] * u0[r] * u0[c]; }
          }
          const double w2 = coe * coe * c2 * c2;
          int t = 0;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int k = r; k < 6; k++) sacc[(t++) * W + i] += w2 * S6[r][k];
        }
        {       // At LAST (round 5): its 18 values are not alive across the noise products above (50 -> 40 spilled registers)
        // At = [(2/NN) A u1, (2/NN) A u2, -(2/NN^2) w]: A u_k = [m0 x u_k + m_k x u0 ; s0 u_k + s_k u0],
          // m_k = (P' - b vbar^T) u_k, s_k = (b - N vbar).u_k, w = [b x u0 ; N u0]   (BAs_left.hpp:418-428,445-446)
          double m0[3], m1[3], m2[3], cvec[3];
          const double vu1 = vbar[0] * u1[0] + vbar[1] * u1[1] + vbar[2] * u1[2];
          const double vu2 = vbar[0] * u2[0] + vbar[1] * u2[1] + vbar[2] * u2[2];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            cvec[r] = b[r] - N * vbar[r];
            m0[r] = Pw[r][0] * u0[0] + Pw[r][1] * u0[1] + Pw[r][2] * u0[2] - b[r] * vu0;
            m1[r] = Pw[r][0] * u1[0] + Pw[r][1] * u1[1] + Pw[r][2] * u1[2] - b[r] * vu1;
            m2[r] = Pw[r][0] * u2[0] + Pw[r][1] * u2[1] + Pw[r][2] * u2[2] - b[r] * vu2;
          }
          const double s0 = cvec[0] * u0[0] + cvec[1] * u0[1] + cvec[2] * u0[2];
          const double s1 = cvec[0] * u1[0] + cvec[1] * u1[1] + cvec[2] * u1[2];
          const double s2 = cvec[0] * u2[0] + cvec[1] * u2[1] + cvec[2] * u2[2];
          double x01[3], x10[3], x02[3], x20[3], bxu[3];
          cross3(m0, u1, x01); cross3(m1, u0, x10);
          cross3(m0, u2, x02); cross3(m2, u0, x20);
          cross3(b, u0, bxu);
          const double c3 = -2.0 * iNN * iNN;
#pragma unroll
          for (int r = 0; r < 3; r++) {
            at[r][0] = c2 * (x01[r] + x10[r]); at[3 + r][0] = c2 * (s0 * u1[r] + s1 * u0[r]);
            at[r][1] = c2 * (x02[r] + x20[r]); at[3 + r][1] = c2 * (s0 * u2[r] + s2 * u0[r]);
            at[r][2] = c3 * bxu[r];            at[3 + r][2] = c3 * N * u0[r];
          }
        }
      }
      if (!ONEPASS) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
          for (int r = 0; r < 6; r += 2) {
            *reinterpret_cast<d2 *>(gx + (size_t)c * npad + 6 * i + r) = (d2){at[r][c], at[r + 1][c]};
            *reinterpret_cast<d2 *>(gy + (size_t)c * npad + 6 * i + r) = (d2){rr[r][c], rr[r + 1][c]};
          }
      }
    }
    for (int r = 6 * W + threadIdx.x; r < npad; r += blockDim.x)
#pragma unroll
      for (int c = 0; c < 3; c++) { gx[(size_t)c * npad + r] = 0.0; gy[(size_t)c * npad + r] = 0.0; }

    // Q = sum over the feature's poses, fixed order: lanes of a wave, then waves
#pragma unroll
    for (int t = 0; t < 6; t++) {
      double s = q[t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) sq[wv][t] = s;
    }
    __syncthreads();
    double Q[6];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      double s = sq[0][t];
      for (int w = 1; w < nwv; w++) s += sq[w][t];
      Q[t] = s;
    }
    // Q = Cq Cq^T (lower), Ci = Cq^-1; a vanished pivot (degenerate feature) drops its column: pseudo-inverse
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Ci[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    {
      const double Qm[3][3] = {{Q[0], Q[1], Q[2]}, {Q[1], Q[3], Q[4]}, {Q[2], Q[4], Q[5]}};
      double d = Qm[0][0];
      if (d > 0) { C[0][0] = sqrt(d); C[1][0] = Qm[1][0] / C[0][0]; C[2][0] = Qm[2][0] / C[0][0]; }
      d = Qm[1][1] - C[1][0] * C[1][0];
      if (Qm[1][1] > 0 && d > 1e-12 * Qm[1][1]) { C[1][1] = sqrt(d); C[2][1] = (Qm[2][1] - C[2][0] * C[1][0]) / C[1][1]; }
      d = Qm[2][2] - C[2][0] * C[2][0] - C[2][1] * C[2][1];
      if (Qm[2][2] > 0 && d > 1e-12 * Qm[2][2]) C[2][2] = sqrt(d);
      // inverse of the live principal part (unit rows/columns for dropped pivots are left zero)
      const double i0 = C[0][0] > 0 ? 1.0 / C[0][0] : 0.0, i1 = C[1][1] > 0 ? 1.0 / C[1][1] : 0.0,
                   i2 = C[2][2] > 0 ? 1.0 / C[2][2] : 0.0;
      Ci[0][0] = i0; Ci[1][1] = i1; Ci[2][2] = i2;
      Ci[1][0] = -C[1][0] * i0 * i1;
      Ci[2][1] = -C[2][1] * i1 * i2;
      Ci[2][0] = -(C[2][0] * Ci[0][0] + C[2][1] * Ci[1][0]) * i2;
    }
    // phase 2: X = coe (At Cq + Y'), Y = coe Y', Y' = Rr Cq^-T
    if (ONEPASS) {
      // from the registers; a wavefront's block of a column (6 x its poses inside the window, contiguous) through the staging block
      const int i0 = (int)threadIdx.x - lane;                   // the wavefront's first pose
      if (i0 < W) {
        const int nval = 6 * min(64, W - i0);
        double *stg = sm + (12 + COV_DACC) * W + wv * 384;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double xc[6], yc[6];
#pragma unroll
          for (int r = 0; r < 6; r++) {
            const double y = rr[r][0] * Ci[c][0] + rr[r][1] * Ci[c][1] + rr[r][2] * Ci[c][2];
            const double x = at[r][0] * C[0][c] + at[r][1] * C[1][c] + at[r][2] * C[2][c];
            xc[r] = coe * (x + y);
            yc[r] = coe * y;
          }
          auto flush = [&](const double col[6], double *gcol) {
            d2 *q2 = reinterpret_cast<d2 *>(stg + 6 * lane);
            q2[0] = (d2){col[0], col[1]}; q2[1] = (d2){col[2], col[3]}; q2[2] = (d2){col[4], col[5]};
            asm volatile("" ::: "memory");       // (one wavefront: its LDS operations are performed in order)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const int idx = 2 * (64 * j + lane);
              const d2 w = *reinterpret_cast<const d2 *>(stg + idx);
              if (idx < nval) *reinterpret_cast<d2 *>(gcol + idx) = w;
            }
            asm volatile("" ::: "memory");
          };
          flush(xc, gx + (size_t)c * npad + 6 * i0);
          flush(yc, gy + (size_t)c * npad + 6 * i0);
        }
      }
    } else
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double A[3], Rw[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { A[c] = gx[(size_t)c * npad + 6 * i + r]; Rw[c] = gy[(size_t)c * npad + 6 * i + r]; }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          // (Rr Ci^T)[c] = sum_k Rr[k] Ci[c][k] ; (At Cq)[c] = sum_k At[k] Cq[k][c]
          const double y = Rw[0] * Ci[c][0] + Rw[1] * Ci[c][1] + Rw[2] * Ci[c][2];
          const double x = A[0] * C[0][c] + A[1] * C[1][c] + A[2] * C[2][c];
          gx[(size_t)c * npad + 6 * i + r] = coe * (x + y);
          gy[(size_t)c * npad + 6 * i + r] = coe * y;
        }
      }
    }
    __syncthreads();      // sq is reused by the next feature
  }
  double *dp = dpart + (size_t)blockIdx.x * COV_DACC * W;
  for (int t = threadIdx.x; t < COV_DACC * W; t += blockDim.x) dp[t] = sacc[t];
}

// split-K partial tiles -> one tile set, fixed order
__global__ __launch_bounds__(256) void k_cov_reduce_tiles(const double *__restrict__ part, int SG, long tile_total,
                                                          double *__restrict__ red) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < tile_total; t += (long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int g = 0; g < SG; g++) s += part[(size_t)g * tile_total + t];
    red[t] = s;
  }
}

__global__ __launch_bounds__(256) void k_cov_reduce_dacc(const double *__restrict__ dpart, int nblk, int len,
                                                         double *__restrict__ out) {
  // 64 outputs per workgroup, the block range split over its four waves, eight loads in flight; fixed order
  __shared__ double sq[256];
  const int jl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  const int chunk = (nblk + 3) / 4, b0 = q * chunk, b1 = min(nblk, b0 + chunk);
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (j < len) {
    int b = b0;
    for (; b + 8 <= b1; b += 8)
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] += dpart[(size_t)(b + k) * len + j];
    for (; b < b1; b++) a[0] += dpart[(size_t)b * len + j];
  }
  sq[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (q == 0 && j < len) out[j] = (sq[jl] + sq[64 + jl]) + (sq[128 + jl] + sq[192 + jl]);
}

// Rcov_raw = XX^T - YY^T + blockdiag(S), both triangles.  MFMA f64 16x16x4 C/D layout of the tile sets:
// col = lane & 15, row = (lane >> 4) + 4 * reg (see k_assemble).
__global__ __launch_bounds__(256) void k_cov_assemble(const double *__restrict__ redx, const double *__restrict__ redy,
                                                      const double *__restrict__ sdiag, const int *__restrict__ tileIJ,
                                                      int ntiles, int W, double *__restrict__ Rout) {
  const int n = 6 * W;
  const long total = (long)ntiles * TILE_ELEMS;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int tile = (int)(t / TILE_ELEMS);
    const int e = (int)(t - (long)tile * TILE_ELEMS);
    const int lane = e & 63, slot = e >> 6;
    const int reg = slot & 3, mt = slot >> 2;
    const int rc = tileIJ[tile * 25 + mt];
    const int row = (rc >> 16) * 16 + (lane >> 4) + 4 * reg;
    const int col = (rc & 0xffff) * 16 + (lane & 15);
    if (row >= n || col >= n || row > col) continue;
    double val = redx[t] - redy[t];
    const int pi = row / 6, pj = col / 6;
    if (pi == pj) {
      const int r = row - 6 * pi, c = col - 6 * pi;         // r <= c
      val += sdiag[(size_t)(r * 6 - r * (r - 1) / 2 + (c - r)) * W + pi];
    }
    Rout[(size_t)col * n + row] = val;
    Rout[(size_t)row * n + col] = val;
  }
}

// ------------------------------------------------------------------------------------------------
// Rcov = H^-1 Rraw H^-T without a single triangular solve: launch_solve leaves, next to L and D, the matrix
// M = L^-T D^+ (the identity rows that ride through the factorisation, kernels_solve.hip), and
//     (P H P^T)^-1 = L^-T D^-1 L^-1 = M D M^T ,
// so with Rp = P Rraw P^T:   Rcov_p = M D (M^T Rp M) D M^T  -- four products with the triangular M on the
// FP64 matrix cores (k_tri_gemm: 48 x 48 tiles, one wavefront per 16 x 48 strip, the k range cut to M's non-zero
// part), no panel chain.  All matrices nA x nA, column-major, padded rows/columns are the identity's.
// ------------------------------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));

// Rp[r][c] = R[perm[r]][perm[c]] (zero where a padded index is involved); inverse: R[perm[r]][perm[c]] = Rp[r][c]
__global__ __launch_bounds__(256) void k_sym_permute(const double *__restrict__ src, int n, int nA, const int *__restrict__ perm,
                                                     int inverse, double *__restrict__ dst) {
  const long total = (long)nA * nA;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / nA), r = (int)(t - (long)c * nA);
    const int pr = perm[r], pc = perm[c];
    if (!inverse) dst[t] = (pr < n && pc < n) ? src[(size_t)pc * n + pr] : 0.0;
    else if (pr < n && pc < n) dst[(size_t)pc * n + pr] = src[t];
  }
}

__global__ __launch_bounds__(256) void k_sym_scale(double *__restrict__ T, int nA, const double *__restrict__ dvec) {
  const long total = (long)nA * nA;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / nA), r = (int)(t - (long)c * nA);
    T[t] *= dvec[r] * dvec[c];
  }
}

// C (nA x nA, ld nA) = A B with A(i,k) = a[i sai + k sak], B(k,j) = b[k sbk + j sbj] (one stride of each operand is 1); one of the operands
// is the triangular M or its transpose: tri = 1: A non-zero for k <= i, 2: B non-zero for k <= j, 3: A non-zero for k >= i, 4: B non-zero
// for k >= j.  Workgroup = 3 waves = a 48 x 48 tile of C, wave w = rows 16 w .. 16 w + 15.
// MFMA f64 16x16x4: A operand lane l = A(row l & 15, k + (l >> 4)), B operand lane l = B(k + (l >> 4), col l & 15),
// C/D: col = l & 15, row = (l >> 4) + 4 reg.
// Round 5: both operand blocks of a 48-wide k block go through LDS.  Until then every wavefront loaded its operands straight from memory in
// MFMA layout -- sixteen separate 32-byte pieces per load instruction (strides ldA / nA) and the same B block three times per tile: 576 MB of
// requests per product, 160 us for 25 us of MFMA work, and issuing the next block's loads ahead of the MFMAs changed nothing
// (profiles/r05t_tri_gemm_prefetch.txt: request-bound, not latency-bound).  Now the 192 threads fetch the two 48 x 48 blocks once, along the
// operand's unit stride (48 consecutive doubles per row: whole 128-byte lines), the next block's 24 values per thread in flight under this
// block's 36 MFMAs per wavefront, and the MFMA operands are LDS reads: As[k][i], Bs[k][j], rows of 49 doubles (an odd row keeps the
// transposing stores of a k-contiguous operand at two lanes per bank; the reads -- sixteen consecutive doubles per k -- are conflict-free
// but for one bank pair).
constexpr int TG_LD = NB + 1;
__global__ __launch_bounds__(192) void k_tri_gemm(const double *__restrict__ a, long sai, long sak, const double *__restrict__ b,
                                                  long sbk, long sbj, int nA, int tri, double *__restrict__ C) {
  __shared__ double As[NB * TG_LD], Bs[NB * TG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int i0 = blockIdx.x * NB, j0 = blockIdx.y * NB;
  int k0 = 0, k1 = nA;
  if (tri == 1) k1 = min(nA, i0 + NB);
  else if (tri == 2) k1 = min(nA, j0 + NB);
  else if (tri == 3) k0 = i0;
  else if (tri == 4) k0 = j0;
  // this thread's twelve elements of a 48 x 48 block: one index along the unit stride (f), the other g0, g0 + 4, ..
  const int f = tid % NB, g0 = tid / NB;
  const bool a_i_fast = sai == 1, b_k_fast = sbk == 1;
  // A block: (i, k) = (f, g) or (g, f); B block: (k, j) = (f, g) or (g, f)
  const double *pa = a_i_fast ? a + (size_t)(i0 + f) + (size_t)g0 * sak : a + (size_t)(i0 + g0) * sai + (size_t)f;
  const long sa_m = a_i_fast ? 4 * sak : 4 * sai, sa_k = a_i_fast ? sak : 1;          // per m (g += 4), per unit of k
  const double *pb = b_k_fast ? b + (size_t)f + (size_t)(j0 + g0) * sbj : b + (size_t)g0 * sbk + (size_t)(j0 + f);
  const long sb_m = b_k_fast ? 4 * sbj : 4 * sbk, sb_k = b_k_fast ? 1 : sbk;
  const int la = a_i_fast ? g0 * TG_LD + f : f * TG_LD + g0, la_m = a_i_fast ? 4 * TG_LD : 4;      // As[k][i]
  const int lb = b_k_fast ? f * TG_LD + g0 : g0 * TG_LD + f, lb_m = b_k_fast ? 4 : 4 * TG_LD;      // Bs[k][j]
  d4 acc[3];
#pragma unroll
  for (int y = 0; y < 3; y++) acc[y] = (d4){0.0, 0.0, 0.0, 0.0};
  double ra[12], rb[12];
  if (k0 < k1) {
#pragma unroll
    for (int m = 0; m < 12; m++) { ra[m] = pa[(size_t)k0 * sa_k + m * sa_m]; rb[m] = pb[(size_t)k0 * sb_k + m * sb_m]; }
  }
  for (int kb = k0; kb < k1; kb += NB) {
    __syncthreads();                               // the last block's operand reads are done
#pragma unroll
    for (int m = 0; m < 12; m++) { As[la + m * la_m] = ra[m]; Bs[lb + m * lb_m] = rb[m]; }
    __syncthreads();
    if (kb + NB < k1) {                            // the next block: in flight under this one's MFMAs
#pragma unroll
      for (int m = 0; m < 12; m++) { ra[m] = pa[(size_t)(kb + NB) * sa_k + m * sa_m]; rb[m] = pb[(size_t)(kb + NB) * sb_k + m * sb_m]; }
    }
#pragma unroll
    for (int u = 0; u < 12; u++) {
      const double av = As[(4 * u + l4) * TG_LD + 16 * wv + l15];
#pragma unroll
      for (int y = 0; y < 3; y++)
        acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[(4 * u + l4) * TG_LD + 16 * y + l15], acc[y], 0, 0, 0);
    }
  }
#pragma unroll
  for (int y = 0; y < 3; y++)
#pragma unroll
    for (int e = 0; e < 4; e++)
      C[(size_t)(j0 + 16 * y + l15) * nA + i0 + 16 * wv + l4 + 4 * e] = acc[y][e];
}

inline int grid1(long total, int bs, int cap) {
  long g = (total + bs - 1) / bs;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

static bool cov_onepass(int W) {          // one pose per lane, and room for the staging blocks; BALM_COV_ONEPASS=0: the two-pass kernel (A/B, tests)
  const char *e = getenv("BALM_COV_ONEPASS");
  return W <= 256 && !(e && e[0] == '0');
}
constexpr size_t COV_STAGE_BYTES = 4 * 384 * sizeof(double);

int cov_factors_grid(int W, int F) {
  size_t lds = (size_t)(12 + COV_DACC) * W * sizeof(double) + (cov_onepass(W) ? COV_STAGE_BYTES : 0);
  int per_cu = (int)(150 * 1024 / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int grid = 256 * per_cu;
  if (grid > F) grid = F;
  return grid < 1 ? 1 : grid;
}

// X, Y columns ([3F][npad] each) and per-block S partials of features [0, F) at the poses the eigen records in
// `feat` were computed for
// see prepare_device_accum(): per-device attribute, set once per context by balm_create
hipError_t prepare_device_cov() {
  hipError_t e = hipFuncSetAttribute((const void *)k_cov_factors<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);   // + static sq
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cov_factors<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cov_factors<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_cov_factors<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
  return e;
}

void launch_cov_factors(hipStream_t s, const double *cl, const double *ccov, double sigma2, const double *poses,
                        const double *feat, int W, int npad, int F, double *Gx, double *Gy, double *dpart, int nblk) {
  const bool one = cov_onepass(W);
  size_t lds = (size_t)(12 + COV_DACC) * W * sizeof(double) + (one ? COV_STAGE_BYTES : 0);
  int bs = W <= 64 ? 64 : (W <= 128 ? 128 : 256);
#define BALM_COV_FACTORS(E, O) hipLaunchKernelGGL((k_cov_factors<E, O>), dim3(nblk), dim3(bs), lds, s, cl, ccov, sigma2, poses, feat, W, npad, F, Gx, Gy, dpart)
  if (ccov) { if (one) BALM_COV_FACTORS(true, true); else BALM_COV_FACTORS(true, false); }
  else { if (one) BALM_COV_FACTORS(false, true); else BALM_COV_FACTORS(false, false); }
#undef BALM_COV_FACTORS
}

void launch_cov_reduce_tiles(hipStream_t s, const double *part, int SG, long tile_total, double *red) {
  hipLaunchKernelGGL(k_cov_reduce_tiles, dim3(grid1(tile_total, 256, 8192)), dim3(256), 0, s, part, SG, tile_total, red);
}

void launch_cov_reduce_dacc(hipStream_t s, const double *dpart, int nblk, int W, double *out) {
  hipLaunchKernelGGL(k_cov_reduce_dacc, dim3((COV_DACC * W + 63) / 64), dim3(256), 0, s, dpart, nblk, COV_DACC * W, out);
}

void launch_cov_assemble(hipStream_t s, const double *redx, const double *redy, const double *sdiag, const int *tileIJ,
                         int ntiles, int W, double *Rout) {
  hipLaunchKernelGGL(k_cov_assemble, dim3(grid1((long)ntiles * TILE_ELEMS, 256, 4096)), dim3(256), 0, s, redx, redy, sdiag,
                     tileIJ, ntiles, W, Rout);
}

// Rcov (n x n) = H^-1 Rraw H^-T, H factored in the context by launch_solve (L, D and M = L^-T D^+ in d_A / d_dvec);
// T0, T1 = nA x nA scratch each
void launch_congruence_inverse(balm_ctx *c, const double *Rraw, double *T0, double *T1, double *Rcov) {
  hipStream_t s = c->stream;
  const int n = c->n, nA = c->nA;
  const long ldA = 2L * nA + NB;
  const double *M = c->d_A + (size_t)(nA + NB);           // M(r, c) = M[r + c ldA]
  const dim3 grid(nA / NB, nA / NB), blk(192);
  const int g1 = grid1((long)nA * nA, 256, 4096);
  hipLaunchKernelGGL(k_sym_permute, dim3(g1), dim3(256), 0, s, Rraw, n, nA, c->d_perm, 0, T0);                    // Rp
  hipLaunchKernelGGL(k_tri_gemm, grid, blk, 0, s, M, ldA, 1L, T0, 1L, (long)nA, nA, 1, T1);                       // M^T Rp
  hipLaunchKernelGGL(k_tri_gemm, grid, blk, 0, s, T1, 1L, (long)nA, M, 1L, ldA, nA, 2, T0);                       // (M^T Rp) M
  hipLaunchKernelGGL(k_sym_scale, dim3(g1), dim3(256), 0, s, T0, nA, c->d_dvec);                                  // D . D
  hipLaunchKernelGGL(k_tri_gemm, grid, blk, 0, s, M, 1L, ldA, T0, 1L, (long)nA, nA, 3, T1);                       // M (.)
  hipLaunchKernelGGL(k_tri_gemm, grid, blk, 0, s, T1, 1L, (long)nA, M, ldA, 1L, nA, 4, T0);                       // (.) M^T
  hipLaunchKernelGGL(k_sym_permute, dim3(g1), dim3(256), 0, s, T0, n, nA, c->d_perm, 1, Rcov);
}

}  // namespace balm
