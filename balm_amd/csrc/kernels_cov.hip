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
//   k_trsm_panel    H^-1 B for n right-hand sides through the LDL^T factor of kernels_solve.hip (applied twice)
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

// c_cov of PointCluster::push with p_cov = sigma^2 I (toolss.hpp:321-345) from the cluster's own moments:
// sum_k Bf Bf^T is linear in (P, v, N)
__device__ __forceinline__ void noise_cov_isotropic(const double P[6], const double v[3], double N, double s2,
                                                    double c[9][9]) {
#pragma unroll
  for (int r = 0; r < 9; r++)
#pragma unroll
    for (int k = 0; k < 9; k++) c[r][k] = 0.0;
  const double xx = P[0], xy = P[1], xz = P[2], yy = P[3], yz = P[4], zz = P[5], x = v[0], y = v[1], z = v[2];
#define BALM_SET(r, k, val) c[r][k] = c[k][r] = s2 * (val)
  BALM_SET(0, 0, 4 * xx); BALM_SET(0, 1, 2 * xy); BALM_SET(0, 2, 2 * xz); BALM_SET(0, 6, 2 * x);
  BALM_SET(1, 1, yy + xx); BALM_SET(1, 2, yz); BALM_SET(1, 3, 2 * xy); BALM_SET(1, 4, xz); BALM_SET(1, 6, y); BALM_SET(1, 7, x);
  BALM_SET(2, 2, zz + xx); BALM_SET(2, 4, xy); BALM_SET(2, 5, 2 * xz); BALM_SET(2, 6, z); BALM_SET(2, 8, x);
  BALM_SET(3, 3, 4 * yy); BALM_SET(3, 4, 2 * yz); BALM_SET(3, 7, 2 * y);
  BALM_SET(4, 4, zz + yy); BALM_SET(4, 5, 2 * yz); BALM_SET(4, 7, z); BALM_SET(4, 8, y);
  BALM_SET(5, 5, 4 * zz); BALM_SET(5, 8, 2 * z);
  BALM_SET(6, 6, N); BALM_SET(7, 7, N); BALM_SET(8, 8, N);
#undef BALM_SET
}

constexpr int COV_DACC = 21;      // upper triangle of the 6x6 block S_j, row-major

// ------------------------------------------------------------------------------------------------
// One workgroup per feature at a time, one lane per pose (like k_feature_factors).  Phase 1 computes the
// pose's At / Rr rows (parked in the X / Y columns), its share of Q and of S_j; after the block-wide sum of Q
// every lane factors the 3x3 Q redundantly and turns its own rows into X and Y.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cov_factors(const double *__restrict__ cl, const double *__restrict__ ccov,
                                                     double sigma2, const double *__restrict__ poses,
                                                     const double *__restrict__ feat, int W, int npad, int F,
                                                     double *__restrict__ Gx, double *__restrict__ Gy,
                                                     double *__restrict__ dpart) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *sp = sm;                 // [12][W] poses
  double *sacc = sm + 12 * W;      // [21][W]
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

    for (int i = threadIdx.x; i < W; i += blockDim.x) {
      double at[6][3], rr[6][3];
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) at[r][c] = rr[r][c] = 0.0;
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
        const double c2 = 2.0 * iNN, c3 = -2.0 * iNN * iNN;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          at[r][0] = c2 * (x01[r] + x10[r]); at[3 + r][0] = c2 * (s0 * u1[r] + s1 * u0[r]);
          at[r][1] = c2 * (x02[r] + x20[r]); at[3 + r][1] = c2 * (s0 * u2[r] + s2 * u0[r]);
          at[r][2] = c3 * bxu[r];            at[3 + r][2] = c3 * N * u0[r];
        }
        // Gm (3x9): rows u_k^T Gkl / ((lam0 - lam_k) NN), k = 1, 2, and m = [0 0 0 0 0 0 r3]      (:431-441)
        //   Gkl[:3] = R g1([r3; p.u0])[:3] + (p - vbar) (x) [0..0 r3] - [0 | (vbar.u0) R]
        double r3[3];
#pragma unroll
        for (int c = 0; c < 3; c++) r3[c] = R[3 * c] * u0[0] + R[3 * c + 1] * u0[1] + R[3 * c + 2] * u0[2];   // R^T u0
        const double pu = p[0] * u0[0] + p[1] * u0[1] + p[2] * u0[2];
        double gm[3][9];
        {
          const double wa[4] = {r3[0], r3[1], r3[2], pu};
          double g1a[3][9];
          g1_top(wa, g1a);
          double ru1[3], ru2[3];     // R^T u_k  (u_k^T R)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            ru1[c] = R[3 * c] * u1[0] + R[3 * c + 1] * u1[1] + R[3 * c + 2] * u1[2];
            ru2[c] = R[3 * c] * u2[0] + R[3 * c + 1] * u2[1] + R[3 * c + 2] * u2[2];
          }
          const double pv1 = (p[0] - vbar[0]) * u1[0] + (p[1] - vbar[1]) * u1[1] + (p[2] - vbar[2]) * u1[2];
          const double pv2 = (p[0] - vbar[0]) * u2[0] + (p[1] - vbar[1]) * u2[1] + (p[2] - vbar[2]) * u2[2];
#pragma unroll
          for (int c = 0; c < 9; c++) {
            double e1 = ru1[0] * g1a[0][c] + ru1[1] * g1a[1][c] + ru1[2] * g1a[2][c];
            double e2 = ru2[0] * g1a[0][c] + ru2[1] * g1a[1][c] + ru2[2] * g1a[2][c];
            if (c >= 6) {
              e1 += pv1 * r3[c - 6] - vu0 * ru1[c - 6];
              e2 += pv2 * r3[c - 6] - vu0 * ru2[c - 6];
            }
            gm[0][c] = k1 * e1;
            gm[1][c] = k2 * e2;
            gm[2][c] = c >= 6 ? r3[c - 6] : 0.0;
          }
        }
        // D = (2/NN) U_0 Y (6x9),  Y = T_j g1([r3; (p - vbar).u0]) (4x9), U_0 = [[hat(-u0), 0], [0, u0]]    (:449-450)
        //   Y[:3] = R g1t[:3] + p (x) [0..0 r3],  Y[3] = [0 0 0 0 0 0 r3]
        // everything downstream is done on the 4-row Y and lifted through U_0 at the end
        double Y[3][9];
        {
          const double st = pu - vu0;
          const double wt[4] = {r3[0], r3[1], r3[2], st};
          double g1t[3][9];
          g1_top(wt, g1t);
#pragma unroll
          for (int c = 0; c < 9; c++)
#pragma unroll
            for (int r = 0; r < 3; r++) {
              double y = R[r] * g1t[0][c] + R[3 + r] * g1t[1][c] + R[6 + r] * g1t[2][c];
              if (c >= 6) y += p[r] * r3[c - 6];
              Y[r][c] = y;
            }
        }
        // the cluster's 9x9 noise covariance
        double cc[9][9];
        if (ccov) {
          const double *src = ccov + ((size_t)a * W + i) * 81;
#pragma unroll
          for (int r = 0; r < 9; r++)
#pragma unroll
            for (int k = 0; k < 9; k++) cc[r][k] = src[9 * r + k];
        } else {
          noise_cov_isotropic(P, v, N, sigma2, cc);
        }
        double sg[9][3];                       // c_cov Gm^T  (Gm row 2 = [0 0 0 0 0 0 r3])
#pragma unroll
        for (int r = 0; r < 9; r++) {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 9; c++) s += cc[r][c] * gm[k][c];
            sg[r][k] = s;
          }
          sg[r][2] = cc[r][6] * r3[0] + cc[r][7] * r3[1] + cc[r][8] * r3[2];
        }
        {
          int t = 0;
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = r; k < 3; k++) {
              double s = 0.0;
#pragma unroll
              for (int c = (r == 2 ? 6 : 0); c < 9; c++) s += gm[r][c] * sg[c][k];
              q[t++] += s;
            }
        }
        // Rr = D c_cov Gm^T = c2 U_0 (Y sg):  rows 0..2 = (Y[:3] sg) x u0 per column, rows 3..5 = u0 (Y[3] sg)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          double ys[3];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 9; c++) s += Y[r][c] * sg[c][k];
            ys[r] = s;
          }
          const double y3 = r3[0] * sg[6][k] + r3[1] * sg[7][k] + r3[2] * sg[8][k];
          double yx[3];
          cross3(ys, u0, yx);                  // hat(-u0) y = y x u0
#pragma unroll
          for (int r = 0; r < 3; r++) { rr[r][k] = c2 * yx[r]; rr[3 + r][k] = c2 * u0[r] * y3; }
        }
        // S_j = D c_cov D^T = c2^2 U_0 M U_0^T,  M = Y4 c_cov Y4^T (4x4 symmetric)
        {
          double M[4][4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            double e[9];                       // row r of Y4 c_cov
#pragma unroll
            for (int c = 0; c < 9; c++) {
              double sacc_ = 0.0;
              if (r < 3) {
#pragma unroll
                for (int k = 0; k < 9; k++) sacc_ += Y[r][k] * cc[k][c];
              } else {
                sacc_ = r3[0] * cc[6][c] + r3[1] * cc[7][c] + r3[2] * cc[8][c];
              }
              e[c] = sacc_;
            }
#pragma unroll
            for (int k = r; k < 4; k++) {
              double m = 0.0;
              if (k < 3) {
#pragma unroll
                for (int c = 0; c < 9; c++) m += e[c] * Y[k][c];
              } else {
                m = e[6] * r3[0] + e[7] * r3[1] + e[8] * r3[2];
              }
              M[r][k] = M[k][r] = m;
            }
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
      }
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 6; r += 2) {
          *reinterpret_cast<d2 *>(gx + (size_t)c * npad + 6 * i + r) = (d2){at[r][c], at[r + 1][c]};
          *reinterpret_cast<d2 *>(gy + (size_t)c * npad + 6 * i + r) = (d2){rr[r][c], rr[r + 1][c]};
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
    // phase 2: X = coe (At Cq + Y'), Y = coe Y', Y' = Rr Cq^-T  (each lane re-reads the rows it parked)
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
// H^-1 B for m right-hand sides through the factor P H P^T = L D L^T left in c->d_A / d_dvec / d_perm by
// launch_solve: Z = P B, L Z' = Z (forward, panel by panel), Z'' = D^+ Z', L^T Z''' = Z'' (backward), out = P^T Z'''.
// B is nA x m column-major with leading dimension nA.  Straightforward FP64 FMA kernels, one launch per panel and
// direction (this stage is O(n^3) with n = 6W <= 2880, small next to the SYRKs).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_permute(const double *__restrict__ B, int n, int nA, int m,
                                                      const int *__restrict__ perm, int inverse, double *__restrict__ out) {
  // inverse == 0: out[r][c] = B[perm[r]][c] (B n x m, ld n; out nA x m, ld nA; padded rows zero)
  // inverse == 1: out[perm[r]][c] = B[r][c] (B nA x m; out n x m)
  const long total = (long)nA * m;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t / nA), r = (int)(t - (long)c * nA);
    const int p = perm[r];
    if (!inverse) out[t] = p < n ? B[(size_t)c * n + p] : 0.0;
    else if (p < n) out[(size_t)c * n + p] = B[t];
  }
}

__global__ __launch_bounds__(256) void k_rows_scale(double *__restrict__ Z, int nA, int m, const double *__restrict__ dvec) {
  const long total = (long)nA * m;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int r = (int)(t % nA);
    const double d = dvec[r];
    Z[t] = fabs(d) > DBL_MIN ? Z[t] / d : 0.0;          // Eigen's D^+ rule (as in k_ldl_panel)
  }
}

__global__ __launch_bounds__(256) void k_transpose_sq(const double *__restrict__ A, int n, double *__restrict__ At) {
  __shared__ double tile[16][17];
  const int bx = blockIdx.x * 16, by = blockIdx.y * 16, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if (bx + tx < n && by + ty < n) tile[ty][tx] = A[(size_t)(by + ty) * n + bx + tx];
  __syncthreads();
  if (by + tx < n && bx + ty < n) At[(size_t)(bx + ty) * n + by + tx] = tile[tx][ty];
}

// inverses of the NB x NB unit-lower-triangular diagonal blocks of L, once per factorisation: column j of the
// inverse by forward substitution, one lane per column.  Linv[p][c][r] = (L_pp^-1)(r, c), column-major.
__global__ __launch_bounds__(64) void k_invert_diag_blocks(const double *__restrict__ A, int ldA, double *__restrict__ Linv) {
  __shared__ double Ls[NB][NB + 1];
  const int c0 = blockIdx.x * NB;
  for (int t = threadIdx.x; t < NB * NB; t += 64) {
    const int r = t % NB, c = t / NB;
    Ls[r][c] = r > c ? A[(size_t)(c0 + c) * ldA + c0 + r] : 0.0;
  }
  __syncthreads();
  const int j = threadIdx.x;
  if (j >= NB) return;
  double x[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) x[k] = k == j ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < NB; k++)
#pragma unroll
    for (int i = k + 1; i < NB; i++) x[i] = __builtin_fma(-Ls[i][k], x[k], x[i]);
  double *dst = Linv + ((size_t)blockIdx.x * NB + j) * NB;
#pragma unroll
  for (int k = 0; k < NB; k++) dst[k] = x[k];
}

// One launch per panel: every workgroup (64 right-hand sides x one 64-row tile of the rows still to update) first
// solves the NB x NB unit-triangular diagonal block against its 64 columns of the panel rows -- redundantly per
// row tile, as a product with the block's precomputed inverse (no serial substitution chain on the critical
// path) -- then applies the rank-NB update to its tile:
//   forward   Z[r][:] -= L[r][c0..c0+NB) X      for r >= c0 + NB        (L Z' = Z)
//   backward  Z[r][:] -= L[c0..c0+NB)[r]^T X    for r <  c0             (L^T Z' = Z)
// The solved panel rows X go to a second buffer S (written by the workgroups of row tile 0), so nobody reads rows
// another workgroup is overwriting.  4 x 4 outputs per lane.
template <int BACKWARD>
__global__ __launch_bounds__(256) void k_trsm_panel(const double *__restrict__ A, const double *__restrict__ Linv, int ldA, int c0,
                                                    double *__restrict__ Z, double *__restrict__ S, int nA, int m, int r_begin,
                                                    int r_end) {
  __shared__ double LL[NB][64 + 1];      // first the diagonal block (strictly lower part), then [k][row] multipliers of the tile
  __shared__ double Xt[NB][64 + 1];      // [k][rhs]  panel rows of these right-hand sides
  const int cb = blockIdx.x * 64, r0 = r_begin + blockIdx.y * 64;
  {
    const double *li = Linv + (size_t)(c0 / NB) * NB * NB;
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
      const int r = t % NB, c = t / NB;           // (L_pp^-1)(r, c); the backward pass needs its transpose
      if (!BACKWARD) LL[r][c] = li[t]; else LL[c][r] = li[t];
    }
  }
  for (int t = threadIdx.x; t < NB * 64; t += 256) {
    const int k = t % NB, cl = t / NB;
    Xt[k][cl] = cb + cl < m ? Z[(size_t)(cb + cl) * nA + c0 + k] : 0.0;
  }
  const bool has_rows = r0 < r_end;
  double lpre[NB * 64 / 256];              // this lane's multipliers, in flight during the solve
  if (has_rows) {
#pragma unroll
    for (int q = 0; q < NB * 64 / 256; q++) {
      const int t = threadIdx.x + 256 * q;
      int k, rl;
      if (!BACKWARD) { rl = t & 63; k = t >> 6; } else { k = t % NB; rl = t / NB; }
      const int r = r0 + rl;
      lpre[q] = 0.0;
      if (r < r_end) lpre[q] = BACKWARD ? A[(size_t)r * ldA + c0 + k] : A[(size_t)(c0 + k) * ldA + r];
    }
  }
  __syncthreads();
  {
    // X = M Xt with M = L_pp^-1 (lower) or its transpose (upper): 48 x 64 outputs, 12 per lane
    const int cl = threadIdx.x & 63, kq = threadIdx.x >> 6;          // rows kq, kq + 4, ...
    double xo[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; q++) {
      const int r = kq + 4 * q;
      double acc = 0.0;
      if (!BACKWARD) { for (int k = 0; k <= r; k++) acc = __builtin_fma(LL[r][k], Xt[k][cl], acc); }
      else { for (int k = r; k < NB; k++) acc = __builtin_fma(LL[r][k], Xt[k][cl], acc); }
      xo[q] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB / 4; q++) {
      const int r = kq + 4 * q;
      Xt[r][cl] = xo[q];
      if (blockIdx.y == 0 && cb + cl < m) S[(size_t)(cb + cl) * nA + c0 + r] = xo[q];
    }
  }
  if (!has_rows) return;
  __syncthreads();                         // the diagonal block is no longer needed
#pragma unroll
  for (int q = 0; q < NB * 64 / 256; q++) {
    const int t = threadIdx.x + 256 * q;
    int k, rl;
    if (!BACKWARD) { rl = t & 63; k = t >> 6; } else { k = t % NB; rl = t / NB; }
    LL[k][rl] = lpre[q];
  }
  __syncthreads();
  const int tr = (threadIdx.x & 15) * 4, tc = (threadIdx.x >> 4) * 4;
  double acc[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
  for (int k = 0; k < NB; k++) {
    double l[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { l[u] = LL[k][tr + u]; x[u] = Xt[k][tc + u]; }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int w = 0; w < 4; w++) acc[u][w] = __builtin_fma(l[u], x[w], acc[u][w]);
  }
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + tr + u, c = cb + tc + w;
      if (r < r_end && c < m) Z[(size_t)c * nA + r] -= acc[u][w];
    }
}

inline int grid1(long total, int bs, int cap) {
  long g = (total + bs - 1) / bs;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

// out (n x m, ld n) = H^-1 B (n x m, ld n); Z, S = nA x m scratch each
void solve_multi(balm_ctx *c, const double *Linv, const double *B, int m, double *Z, double *S, double *out) {
  hipStream_t s = c->stream;
  const int n = c->n, nA = c->nA, ldA = 2 * nA + NB, P = nA / NB;
  const unsigned gx = (unsigned)((m + 63) / 64);
  hipLaunchKernelGGL(k_rows_permute, dim3(grid1((long)nA * m, 256, 4096)), dim3(256), 0, s, B, n, nA, m, c->d_perm, 0, Z);
  for (int p = 0; p < P; p++) {                 // forward: work in Z, solved rows into S
    const int c0 = p * NB, rows = nA - c0 - NB;
    hipLaunchKernelGGL(k_trsm_panel<0>, dim3(gx, rows > 0 ? (rows + 63) / 64 : 1), dim3(256), 0, s, c->d_A, Linv, ldA, c0, Z, S, nA, m,
                       c0 + NB, nA);
  }
  hipLaunchKernelGGL(k_rows_scale, dim3(grid1((long)nA * m, 256, 4096)), dim3(256), 0, s, S, nA, m, c->d_dvec);
  for (int p = P - 1; p >= 0; p--) {            // backward: work in S, solved rows into Z
    const int c0 = p * NB;
    hipLaunchKernelGGL(k_trsm_panel<1>, dim3(gx, c0 > 0 ? (c0 + 63) / 64 : 1), dim3(256), 0, s, c->d_A, Linv, ldA, c0, S, Z, nA, m, 0,
                       c0);
  }
  hipLaunchKernelGGL(k_rows_permute, dim3(grid1((long)nA * m, 256, 4096)), dim3(256), 0, s, Z, n, nA, m, c->d_perm, 1, out);
}

}  // namespace

int cov_factors_grid(int W, int F) {
  size_t lds = (size_t)(12 + COV_DACC) * W * sizeof(double);
  int per_cu = (int)(150 * 1024 / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int grid = 256 * per_cu;
  if (grid > F) grid = F;
  return grid < 1 ? 1 : grid;
}

// X, Y columns ([3F][npad] each) and per-block S partials of features [0, F) at the poses the eigen records in
// `feat` were computed for
void launch_cov_factors(hipStream_t s, const double *cl, const double *ccov, double sigma2, const double *poses,
                        const double *feat, int W, int npad, int F, double *Gx, double *Gy, double *dpart, int nblk) {
  size_t lds = (size_t)(12 + COV_DACC) * W * sizeof(double);
  int bs = W <= 64 ? 64 : (W <= 128 ? 128 : 256);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void *)k_cov_factors, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);   // + static sq
    attr_set = true;
  }
  hipLaunchKernelGGL(k_cov_factors, dim3(nblk), dim3(bs), lds, s, cl, ccov, sigma2, poses, feat, W, npad, F, Gx, Gy, dpart);
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

// Rcov (n x n) = H^-1 Rraw H^-T, H factored in the context by launch_solve; Z, S (nA x n each), tmp (n x n) and
// Linv (nA x NB) scratch
void launch_congruence_inverse(balm_ctx *c, const double *Rraw, double *Z, double *S, double *tmp, double *Linv, double *Rcov) {
  const int n = c->n, nA = c->nA;
  hipLaunchKernelGGL(k_invert_diag_blocks, dim3(nA / NB), dim3(64), 0, c->stream, c->d_A, 2 * nA + NB, Linv);
  solve_multi(c, Linv, Rraw, n, Z, S, tmp);              // M1 = H^-1 Rraw
  hipLaunchKernelGGL(k_transpose_sq, dim3((n + 15) / 16, (n + 15) / 16), dim3(256), 0, c->stream, tmp, n, Rcov);
  solve_multi(c, Linv, Rcov, n, Z, S, tmp);              // H^-1 M1^T = H^-1 Rraw H^-T  (symmetric)
  hipMemcpyAsync(Rcov, tmp, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
}

}  // namespace balm
