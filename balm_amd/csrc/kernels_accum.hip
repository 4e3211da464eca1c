// Hessian / gradient / residual accumulation kernels for gfx950 (MI355X).  FP64 throughout.
//
// The reference evaluators (src/benchmark/bavoxel.hpp:304-426 left form, :53-158 right form,
// :428-470 residual) are restated through the exact identity
//        Hess = blockdiag_i(B_i) - Gt * Gt^T ,   Gt in R^{6W x 3F}
// (three scaled 6-vectors per (feature, pose)), so the O(F W^2)
// pair loop (:404-418) becomes one FP64 SYRK on the matrix cores:
//   K1  world_moments   one wavefront per feature, coalesced SoA reads, shuffle reduction
//   K1b feature_eigen   one lane per feature: 3x3 Jacobi, residual partials
//   K2  feature_factors one lane per (feature, pose): Gt columns + gradient + B_i partials
//   K3  hessian_syrk    25 MFMA sub-tiles (an 80x80 tile, or 25 upper sub-tiles of the diagonal blocks) per
//                       wavefront on v_mfma_f64_16x16x4_f64, split-K
//   K4  reduce / assemble
#include "balm_internal.h"

namespace balm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct Obs {          // one observation (feature a, pose i) moved to the world frame
  double N;
  double Rv[3];       // R v
  double b[3];        // R v + N p                 (world first moment; tools.hpp:336)
  double Pw[6];       // world second moment P' (xx xy xz yy yz zz; tools.hpp:337-338)
};

// tools.hpp:333-339 (PointCluster::transform) == TCT_i of bavoxel.hpp:334-335
__device__ __forceinline__ void to_world(const double P[6], const double v[3], double N, const double R[9],
                                         const double p[3], Obs &o) {
  // R is column-major: R(r,c) = R[3*c+r]
  o.N = N;
#pragma unroll
  for (int r = 0; r < 3; r++) o.Rv[r] = R[r] * v[0] + R[3 + r] * v[1] + R[6 + r] * v[2];
#pragma unroll
  for (int r = 0; r < 3; r++) o.b[r] = o.Rv[r] + N * p[r];
  const double Pf[3][3] = {{P[0], P[1], P[2]}, {P[1], P[3], P[4]}, {P[2], P[4], P[5]}};
  double RP[3][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) RP[r][c] = R[r] * Pf[0][c] + R[3 + r] * Pf[1][c] + R[6 + r] * Pf[2][c];
  int k = 0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = r; c < 3; c++) {
      double rprt = RP[r][0] * R[c] + RP[r][1] * R[3 + c] + RP[r][2] * R[6 + c];
      o.Pw[k++] = rprt + o.Rv[r] * p[c] + p[r] * o.b[c];
    }
}

// ------------------------------------------------------------------------------------------------
// K0: caller layout [F][W][10] -> per-feature SoA [F][10][W] (one-off, at balm_set_features)
// ------------------------------------------------------------------------------------------------
__global__ void k_transpose_clusters(const double *__restrict__ aos, double *__restrict__ soa, int F, int W) {
  const size_t total = (size_t)F * W;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t a = t / W;
    const int i = (int)(t - a * W);
    const double *src = aos + t * 10;
    double *dst = soa + a * 10 * W + i;
#pragma unroll
    for (int c = 0; c < 10; c++) dst[(size_t)c * W] = src[c];
  }
}

// which (feature, pose) pairs are observed (N != 0), one byte each, from the per-feature SoA table: all the host needs of a
// feature table that was built on the device (planes per pose, work model, block-sparse plan) -- F x W bytes instead of the
// F x W x 80 bytes of the table itself
__global__ void k_obs_mask(const double *__restrict__ soa, int F, int W, unsigned char *__restrict__ mask) {
  const size_t total = (size_t)F * W;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t a = t / W;
    const int i = (int)(t - a * W);
    mask[t] = soa[a * 10 * W + (size_t)9 * W + i] != 0.0 ? 1 : 0;
  }
}

void launch_obs_mask(hipStream_t s, const double *soa, int F, int W, unsigned char *mask) {
  const size_t total = (size_t)F * W;
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_obs_mask, dim3(grid), dim3(256), 0, s, soa, F, W, mask);
}

void launch_transpose_clusters(hipStream_t s, const double *aos, double *soa, int F, int W) {
  size_t total = (size_t)F * W;
  int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_transpose_clusters, dim3(grid), dim3(256), 0, s, aos, soa, F, W);
}

// ------------------------------------------------------------------------------------------------
// K1: per-feature world moments  C_a = sum_i T_i Co_i T_i^T   (bavoxel.hpp:331-339, :443-447)
// One wavefront per feature; lanes stride the W poses; ten coalesced f64 streams per feature.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_world_moments(const double *__restrict__ cl,
                                                       const double *__restrict__ poses, int W, int f0, int f1,
                                                       double *__restrict__ Cout) {
  extern __shared__ __attribute__((aligned(16))) double sp[];   // [12][W]
  for (int t0 = 0; t0 < 12 * W; t0 += 10 * (int)blockDim.x) {     // ten loads per lane in flight at a time (k_feature_factors: why)
    double pv[10];
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int t = t0 + j * (int)blockDim.x + (int)threadIdx.x;
      pv[j] = t < 12 * W ? poses[t] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int t = t0 + j * (int)blockDim.x + (int)threadIdx.x;
      if (t < 12 * W) { const int i = t / 12, c = t - 12 * i; sp[c * W + i] = pv[j]; }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int a = f0 + blockIdx.x * nw + wv; a < f1; a += gridDim.x * nw) {
    const double *ca = cl + (size_t)a * 10 * W;
    double acc[10];
#pragma unroll
    for (int c = 0; c < 10; c++) acc[c] = 0.0;
    for (int i = lane; i < W; i += 64) {
      // all ten streams are loaded unconditionally: one memory latency per observation, not two
      double P[6], v[3];
      const double N = ca[(size_t)9 * W + i];
#pragma unroll
      for (int c = 0; c < 6; c++) P[c] = ca[(size_t)c * W + i];
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = ca[(size_t)(6 + c) * W + i];
      if ((int)N > 0) {
        double R[9], p[3];
#pragma unroll
        for (int c = 0; c < 9; c++) R[c] = sp[c * W + i];
#pragma unroll
        for (int c = 0; c < 3; c++) p[c] = sp[(9 + c) * W + i];
        Obs o;
        to_world(P, v, N, R, p, o);
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] += o.Pw[c];
#pragma unroll
        for (int c = 0; c < 3; c++) acc[6 + c] += o.b[c];
        acc[9] += N;
      }
    }
#pragma unroll
    for (int c = 0; c < 10; c++) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 10; c++) Cout[(size_t)a * 10 + c] = acc[c];
    }
  }
}

void launch_world_moments(hipStream_t s, const double *cl, const double *poses, int W, int f0, int f1, double *C) {
  int nf = f1 - f0;
  if (nf <= 0) return;
  int grid = (nf + 3) / 4;
  if (grid > 2048) grid = 2048;
  size_t lds = (size_t)12 * W * sizeof(double);
  hipLaunchKernelGGL(k_world_moments, dim3(grid), dim3(256), lds, s, cl, poses, W, f0, f1, C);
}

// ------------------------------------------------------------------------------------------------
// K1b: per-feature 3x3 symmetric eigen-decomposition (stand-in for Eigen::SelfAdjointEigenSolver,
// bavoxel.hpp:345-351), residual partials coe*lambda_0 (:349), and the per-feature scale factors
// of the three Gt columns.  One lane per feature.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void jrot(double &app, double &aqq, double &apq, double &arp, double &arq, double &v0p,
                                     double &v0q, double &v1p, double &v1q, double &v2p, double &v2q) {
  if (apq == 0.0) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  app -= t * apq;
  aqq += t * apq;
  apq = 0.0;
  const double rp = c * arp - s * arq, rq = s * arp + c * arq;
  arp = rp; arq = rq;
  double x, y;
  x = c * v0p - s * v0q; y = s * v0p + c * v0q; v0p = x; v0q = y;
  x = c * v1p - s * v1q; y = s * v1p + c * v1q; v1p = x; v1q = y;
  x = c * v2p - s * v2q; y = s * v2p + c * v2q; v2p = x; v2q = y;
}

// eigenvalues ascending in lam[], eigenvector k = (U[0][k], U[1][k], U[2][k])
__device__ __forceinline__ void eig3_jacobi(double a00, double a01, double a02, double a11, double a12, double a22,
                                            double lam[3], double U[3][3]) {
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  for (int sweep = 0; sweep < 30; sweep++) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12;
    const double dia = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    jrot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);   // (p,q)=(0,1), r=2
    jrot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);   // (0,2), r=1
    jrot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);   // (1,2), r=0
  }
  double l0 = a00, l1 = a11, l2 = a22;
  double c0[3] = {v00, v10, v20}, c1[3] = {v01, v11, v21}, c2[3] = {v02, v12, v22};
#define BALM_CSWAP(la, lb, ca, cb)                                                         \
  if (la > lb) {                                                                            \
    double t_ = la; la = lb; lb = t_;                                                       \
    for (int k_ = 0; k_ < 3; k_++) { double s_ = ca[k_]; ca[k_] = cb[k_]; cb[k_] = s_; }    \
  }
  BALM_CSWAP(l0, l1, c0, c1)
  BALM_CSWAP(l1, l2, c1, c2)
  BALM_CSWAP(l0, l1, c0, c1)
#undef BALM_CSWAP
  lam[0] = l0; lam[1] = l1; lam[2] = l2;
  for (int k = 0; k < 3; k++) { U[k][0] = c0[k]; U[k][1] = c1[k]; U[k][2] = c2[k]; }
}

// mail.host != NULL (the LM loop's trial residual, no collective transport, at most 256 features = ONE workgroup): the workgroup puts its
// sum -- the whole residual -- into scal[mail.slot] and sends the iteration's 16 scalars and the stamp to the pinned host mirror itself: one
// launch (k_scalars_mail, ~6 us) less per LM iteration of a small window.  (With more workgroups the last one would have to be found by
// tickets: measured, 0.02 ms per launch at 196 workgroups for the device-scope release each of them needs -- see the note at the end of this file.)
__global__ __launch_bounds__(256) void k_feature_eigen(const double *__restrict__ C, const double *__restrict__ fix,
                                                       const double *__restrict__ coe, int f0, int f1,
                                                       double *__restrict__ feat, double *__restrict__ rpart, EigenMail mail) {
  __shared__ double sred[256];
  const int a = f0 + blockIdx.x * blockDim.x + threadIdx.x;
  double res = 0.0;
  if (a < f1) {
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; k++) c[k] = C[(size_t)a * 10 + k];
    if (fix) {
#pragma unroll
      for (int k = 0; k < 10; k++) c[k] += fix[(size_t)a * 10 + k];
    }
    const double NN = c[9];
    const double inv = 1.0 / NN;
    const double vb0 = c[6] * inv, vb1 = c[7] * inv, vb2 = c[8] * inv;
    double lam[3], U[3][3];
    eig3_jacobi(c[0] * inv - vb0 * vb0, c[1] * inv - vb0 * vb1, c[2] * inv - vb0 * vb2, c[3] * inv - vb1 * vb1,
                c[4] * inv - vb1 * vb2, c[5] * inv - vb2 * vb2, lam, U);
    const double w = coe[a];
    res = w * lam[0];
    double *f = feat + (size_t)a * FEAT_STRIDE;
    f[FT_NN] = NN;
    f[FT_VBAR] = vb0; f[FT_VBAR + 1] = vb1; f[FT_VBAR + 2] = vb2;
    f[FT_LAM] = lam[0]; f[FT_LAM + 1] = lam[1]; f[FT_LAM + 2] = lam[2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      f[FT_U0 + k] = U[k][0]; f[FT_U1 + k] = U[k][1]; f[FT_U2 + k] = U[k][2];
    }
    // column scales of Gt: sqrt(2 coe)/NN, sqrt(2 coe/(lam_k - lam_0))/NN
    f[FT_C0] = sqrt(2.0 * w) * inv;
    f[FT_C1] = sqrt(2.0 * w / (lam[1] - lam[0])) * inv;
    f[FT_C2] = sqrt(2.0 * w / (lam[2] - lam[0])) * inv;
    f[FT_COE] = w;
  }
  sred[threadIdx.x] = res;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sred[threadIdx.x] += sred[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) rpart[blockIdx.x] = sred[0];
  if (!mail.host) return;                                         // (only ever set on a ONE-workgroup launch: sred[0] is the whole sum)
  // What the stamp promises the host is the sixteen scalars, nothing else: the feature records and rpart this kernel wrote above are
  // fenced by the sixteen mailing lanes only, so whatever reads THEM must be ordered behind this kernel on ctx->stream (every reader in
  // the library is: the next kernels of the iteration) -- a host action taken on seeing the stamp must not read them off-stream.
  if (threadIdx.x == 0) mail.scal[mail.slot] = sred[0];
  __syncthreads();
  volatile double *host = mail.host;
  if (threadIdx.x < 16) host[threadIdx.x] = ((int)threadIdx.x == mail.slot) ? sred[0] : mail.scal[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { host[SCAL_STAMP] = mail.stamp; __threadfence_system(); }
}

int launch_feature_eigen(hipStream_t s, const double *C, const double *fix, const double *coe, int f0, int f1,
                         double *feat, double *rpart, const EigenMail *mail) {
  int nf = f1 - f0;
  if (nf <= 0) return 0;
  int grid = (nf + 255) / 256;
  const EigenMail none{nullptr, nullptr, 0.0, 0};
  hipLaunchKernelGGL(k_feature_eigen, dim3(grid), dim3(256), 0, s, C, fix, coe, f0, f1, feat, rpart, (mail && grid == 1) ? *mail : none);
  return grid;
}

// deterministic single-block sum of `nr` partials -> out[0]
__global__ __launch_bounds__(256) void k_sum_scalar(const double *__restrict__ part, int nr, double *__restrict__ out) {
  __shared__ double sred[256];
  double s = 0.0;
  for (int t = threadIdx.x; t < nr; t += 256) s += part[t];
  sred[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sred[threadIdx.x] += sred[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sred[0];
}

void launch_sum_scalar(hipStream_t s, const double *rpart, int nr, double *out) {
  hipLaunchKernelGGL(k_sum_scalar, dim3(1), dim3(256), 0, s, rpart, nr, out);
}

// ------------------------------------------------------------------------------------------------
// K2: per-(feature, pose) factors.  Block = 256 lanes = 256 poses of ONE feature at a time (the
// feature record is wave-uniform -> scalar loads); a lane owns its pose's accumulators in LDS
// ([DACC][W], conflict-free, no atomics).  Writes the three Gt columns of the feature (k-major:
// Gt[(3a+k)*npad + 6i + r], 48 contiguous bytes per lane -> fully coalesced).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ void store6(double *dst, const double x[6]) {
  d2 *q = reinterpret_cast<d2 *>(dst);     // 6i doubles -> 48-byte offsets: 16-byte aligned
  d2 t0 = {x[0], x[1]}, t1 = {x[2], x[3]}, t2 = {x[4], x[5]};
  q[0] = t0; q[1] = t1; q[2] = t2;
}

// One observation (feature a, pose i) of K2: the three Gt columns of the pose, and its gradient / block-diagonal terms added to
// the lane's accumulators in LDS (sacc[k * Wc + il]).  Shared by k_feature_factors and the fused k_moments_factors.
struct FeatRec { double NN, iNN, vbar[3], u0[3], u1[3], u2[3], c0, c1, c2, coe; };

template <int FORM>
__device__ __forceinline__ void obs_factors(const FeatRec &fr, const double P[6], const double v[3], const double N,
                                            const double *__restrict__ sp, double *__restrict__ sacc, const int Wc, const int il,
                                            double col0[6], double col1[6], double col2[6]) {
  const double NN = fr.NN, iNN = fr.iNN, c0 = fr.c0, c1 = fr.c1, c2 = fr.c2, coe = fr.coe;
  const double *vbar = fr.vbar, *u0 = fr.u0, *u1 = fr.u1, *u2 = fr.u2;
  if ((int)N > 0) {
    double R[9], p[3];
#pragma unroll
    for (int c = 0; c < 9; c++) R[c] = sp[c * Wc + il];
#pragma unroll
    for (int c = 0; c < 3; c++) p[c] = sp[(9 + c) * Wc + il];

    if (FORM == 0) {
      // ---- LEFT form, bavoxel.hpp:365-402 -------------------------------------------------
      Obs o;
      to_world(P, v, N, R, p, o);
      // M = TC_i [R_i, p_i - vbar]^T (:368-370):  M_top = P' - b vbar^T ,  M_bot = (b - N vbar)^T
      const double Pw[3][3] = {{o.Pw[0], o.Pw[1], o.Pw[2]}, {o.Pw[1], o.Pw[3], o.Pw[4]}, {o.Pw[2], o.Pw[4], o.Pw[5]}};
      double cvec[3];
#pragma unroll
      for (int r = 0; r < 3; r++) cvec[r] = o.b[r] - N * vbar[r];
      double m0[3], m1[3], m2[3];
      const double vu0 = vbar[0] * u0[0] + vbar[1] * u0[1] + vbar[2] * u0[2];
      const double vu1 = vbar[0] * u1[0] + vbar[1] * u1[1] + vbar[2] * u1[2];
      const double vu2 = vbar[0] * u2[0] + vbar[1] * u2[1] + vbar[2] * u2[2];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        m0[r] = Pw[r][0] * u0[0] + Pw[r][1] * u0[1] + Pw[r][2] * u0[2] - o.b[r] * vu0;
        m1[r] = Pw[r][0] * u1[0] + Pw[r][1] * u1[1] + Pw[r][2] * u1[2] - o.b[r] * vu1;
        m2[r] = Pw[r][0] * u2[0] + Pw[r][1] * u2[1] + Pw[r][2] * u2[2] - o.b[r] * vu2;
      }
      const double s0 = cvec[0] * u0[0] + cvec[1] * u0[1] + cvec[2] * u0[2];
      const double s1 = cvec[0] * u1[0] + cvec[1] * u1[1] + cvec[2] * u1[2];
      const double s2 = cvec[0] * u2[0] + cvec[1] * u2[1] + cvec[2] * u2[2];
      // g_k = (U_k M u_0 + U_0 M u_k)/NN (:371-378) with U_k [x;s] = [x cross u_k ; s u_k]
      double x00[3], x01[3], x10[3], x02[3], x20[3], bxu[3];
      cross3(m0, u0, x00);
      cross3(m0, u1, x01); cross3(m1, u0, x10);
      cross3(m0, u2, x02); cross3(m2, u0, x20);
      cross3(o.b, u0, bxu);                       // w = (U_0 TC_i)[:,3] = [b x u0 ; N u0] (:380)
      double grad[6];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        grad[r] = 2.0 * coe * iNN * x00[r];       // coe * g_0 (:381)
        grad[3 + r] = 2.0 * coe * iNN * s0 * u0[r];
        col0[r] = c0 * bxu[r];
        col0[3 + r] = c0 * N * u0[r];
        col1[r] = c1 * (x01[r] + x10[r]);
        col1[3 + r] = c1 * (s0 * u1[r] + s1 * u0[r]);
        col2[r] = c2 * (x02[r] + x20[r]);
        col2[3 + r] = c2 * (s0 * u2[r] + s2 * u0[r]);
      }
      // B_i = coe (Ell + Ell^T in the top-left 3x3) + (2 coe/NN) U_0 TCT_i U_0^T   (:387-388,:397-402)
      //   Ell + Ell^T = (u0 m0^T + m0 u0^T - 2 (m0.u0) I)/NN
      //   U_0 TCT U_0^T = [[K P' K^T, (b x u0) u0^T],[u0 (b x u0)^T, N u0 u0^T]],  K = hat(u0)
      const double K[3][3] = {{0, -u0[2], u0[1]}, {u0[2], 0, -u0[0]}, {-u0[1], u0[0], 0}};
      double KP[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) KP[r][c] = K[r][0] * Pw[0][c] + K[r][1] * Pw[1][c] + K[r][2] * Pw[2][c];
      const double m0u0 = m0[0] * u0[0] + m0[1] * u0[1] + m0[2] * u0[2];
      const double k1 = coe * iNN, k2 = 2.0 * coe * iNN;
      double bd[21];
      int q = 0;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) {          // TL, symmetric
          double kpk = KP[r][0] * K[c][0] + KP[r][1] * K[c][1] + KP[r][2] * K[c][2];
          double ell = u0[r] * m0[c] + m0[r] * u0[c] - (r == c ? 2.0 * m0u0 : 0.0);
          bd[q++] = k1 * ell + k2 * kpk;
        }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) bd[q++] = k2 * bxu[r] * u0[c];   // TR (rows 0..2, cols 3..5)
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) bd[q++] = k2 * N * u0[r] * u0[c];   // BR, symmetric
#pragma unroll
      for (int k = 0; k < 6; k++) sacc[k * Wc + il] += grad[k];
#pragma unroll
      for (int k = 0; k < 21; k++) sacc[(6 + k) * Wc + il] += bd[k];
    } else {
      // ---- RIGHT form, bavoxel.hpp:93-130 ("Right update.pdf") ----------------------------
      const double Pf[3][3] = {{P[0], P[1], P[2]}, {P[1], P[3], P[4]}, {P[2], P[4], P[5]}};
      const double NNi = (double)(int)NN;            // int NN in the reference (:82)
      const double jNN = 1.0 / NNi;
      double r3[3];                                   // R^T u0
#pragma unroll
      for (int c = 0; c < 3; c++) r3[c] = R[3 * c] * u0[0] + R[3 * c + 1] * u0[1] + R[3 * c + 2] * u0[2];
      double ai[3];
      cross3(v, r3, ai);                              // hat(v) r
      double ti[3];
#pragma unroll
      for (int r = 0; r < 3; r++) ti[r] = p[r] - vbar[r];
      const double s = u0[0] * ti[0] + u0[1] * ti[1] + u0[2] * ti[2];
      double Pr[3];
#pragma unroll
      for (int r = 0; r < 3; r++) Pr[r] = Pf[r][0] * r3[0] + Pf[r][1] * r3[1] + Pf[r][2] * r3[2];
      // combo1 = hat(P r) + hat(v) s ; combo2 = R v + n t
      const double h1[3] = {Pr[0] + v[0] * s, Pr[1] + v[1] * s, Pr[2] + v[2] * s};   // combo1 = hat(h1)
      const double C1[3][3] = {{0, -h1[2], h1[1]}, {h1[2], 0, -h1[0]}, {-h1[1], h1[0], 0}};
      const double rh[3][3] = {{0, -r3[2], r3[1]}, {r3[2], 0, -r3[0]}, {-r3[1], r3[0], 0}};
      double Rv[3], combo2[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        Rv[r] = R[r] * v[0] + R[3 + r] * v[1] + R[6 + r] * v[2];
        combo2[r] = Rv[r] + N * ti[r];
      }
      // Auk = [ ((R P + t v^T) rh - R combo1) , (combo2 u0^T + (combo2.u0) I) ] / NN     (3x6)
      double E[3][3];      // R P + t v^T
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
          E[r][c] = R[r] * Pf[0][c] + R[3 + r] * Pf[1][c] + R[6 + r] * Pf[2][c] + ti[r] * v[c];
      double Auk[3][6];
      const double c2u = combo2[0] * u0[0] + combo2[1] * u0[1] + combo2[2] * u0[2];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double erh = E[r][0] * rh[0][c] + E[r][1] * rh[1][c] + E[r][2] * rh[2][c];
          double rc1 = R[r] * C1[0][c] + R[3 + r] * C1[1][c] + R[6 + r] * C1[2][c];
          Auk[r][c] = (erh - rc1) * jNN;
          Auk[r][3 + c] = (combo2[r] * u0[c] + (r == c ? c2u : 0.0)) * jNN;
        }
      double jjt[6], a1[6], a2[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        jjt[c] = Auk[0][c] * u0[0] + Auk[1][c] * u0[1] + Auk[2][c] * u0[2];
        a1[c] = Auk[0][c] * u1[0] + Auk[1][c] * u1[1] + Auk[2][c] * u1[2];
        a2[c] = Auk[0][c] * u2[0] + Auk[1][c] * u2[1] + Auk[2][c] * u2[2];
      }
      // Gt columns: c0' [a_i ; n u0], c1' Auk^T u1, c2' Auk^T u2 with the 1/NN already in Auk.
      // FT_C* carry 1/NN (double NN); the right form divides by the int NN -> rescale.
      const double r0 = c0 * NN * jNN, r1 = c1 * NN, r2 = c2 * NN;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        col0[r] = r0 * ai[r];
        col0[3 + r] = r0 * N * u0[r];
      }
#pragma unroll
      for (int c = 0; c < 6; c++) { col1[c] = r1 * a1[c]; col2[c] = r2 * a2[c]; }
      // B_i (right) = Hess_ii + (Gt Gt^T)_ii:
      //   TL = coe (2/NN (combo1 - rh P) rh - 0.5 hat(jjt[0:3]))   (general 3x3)
      //   TR = coe 2/NN a_i u0^T ; BR = coe 2 n/NN u0 u0^T
      double D1[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
          D1[r][c] = C1[r][c] - (rh[r][0] * Pf[0][c] + rh[r][1] * Pf[1][c] + rh[r][2] * Pf[2][c]);
      const double hj[3][3] = {{0, -jjt[2], jjt[1]}, {jjt[2], 0, -jjt[0]}, {-jjt[1], jjt[0], 0}};
      const double k2 = 2.0 * coe * jNN;
      double bd[24];
      int q = 0;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          double d1rh = D1[r][0] * rh[0][c] + D1[r][1] * rh[1][c] + D1[r][2] * rh[2][c];
          bd[q++] = k2 * d1rh - 0.5 * coe * hj[r][c];
        }
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) bd[q++] = k2 * ai[r] * u0[c];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = r; c < 3; c++) bd[q++] = k2 * N * u0[r] * u0[c];
#pragma unroll
      for (int k = 0; k < 6; k++) sacc[k * Wc + il] += coe * jjt[k];
#pragma unroll
      for (int k = 0; k < 24; k++) sacc[(6 + k) * Wc + il] += bd[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; k++) col0[k] = col1[k] = col2[k] = 0.0;
  }
}

// REGS (round 4; the left form, one chunk, W <= blockDim: a lane owns ONE pose for the workgroup's whole life): the lane's pose and its DACC
// accumulators live in registers instead of LDS -- no LDS traffic per observation but the staging block's (78 LDS instructions per observation
// before): 0.434 -> 0.408 ms at config 2 on top of the coalesced stores, bit for bit the same sums (profiles/r04l_factors_staged.txt).  (Tried
// in round 1 on the lane-by-lane stores: no gain -- the stores hid it.  The right form's 30 accumulators do not fit beside its working set.)
// MAXR (round 6, BALM_SYRK=int8): the lane -- one pose for the workgroup's whole life -- also keeps the largest |entry| of its six Gt rows and
// leaves it in rowmax_part[workgroup][r][pose]; k_rowmax_reduce takes the maximum over the workgroups: the row exponents of the INT8
// product's digit slicing (kernels_syrk_i8.hip) without a pass of their own over Gt.  (One atomicMax per lane and row on rowmax itself: 512
// workgroups x 1200 addresses, 512 device-scope atomics in a row on every address -- +48 us on the 407 us kernel, measured.)  The default
// instantiations are MAXR = false.
// (fmax drops a NaN: `bad` turns NaN at the first non-finite entry of the pose's rows -- 0 x inf, 0 x NaN -- and stays; such a pose's six maxima are
//  published as NaN, the slicing gives its rows a NaN scale and H gets the non-finite rows and columns the FP64 product would give it)
__device__ __forceinline__ void track_rowmax(double rm[6], double &bad, const double col0[6], const double col1[6], const double col2[6]) {
#pragma unroll
  for (int k = 0; k < 6; k++) {
    rm[k] = fmax(rm[k], fmax(fabs(col0[k]), fmax(fabs(col1[k]), fabs(col2[k]))));
    bad = fma(col0[k] + col1[k] + col2[k], 0.0, bad);
  }
}
__device__ __forceinline__ void publish_rowmax(double *__restrict__ part, int W, int pose, const double rm[6], double bad) {
#pragma unroll
  for (int k = 0; k < 6; k++) part[(size_t)(blockIdx.x * 6 + k) * W + pose] = bad != bad ? bad : rm[k];
}
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? __longlong_as_double(0x7ff8000000000000ll) : fmax(a, b); }
// rowmax[6 pose + r] = max over the workgroups, as the bit pattern the slicing kernel reads (non-negative doubles order like their bits);
// rows beyond the window: 0
__global__ __launch_bounds__(1024) void k_rowmax_reduce(const double *__restrict__ part, int nblk, int W, int npad, unsigned long long *__restrict__ rowmax) {
  // 64 values t = r * W + pose per workgroup (consecutive lanes, consecutive addresses of a workgroup's record), sixteen wavefronts share the
  // records -- 512 of them at config 2, a chain of dependent maxima per value otherwise (one wavefront per 64 values: 44 us)
  __shared__ double sm[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, t = blockIdx.x * 64 + tx;
  double m[4] = {0.0, 0.0, 0.0, 0.0};
  if (t < 6 * W) {
    int b = ty;
    for (; b + 48 < nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 4; u++) m[u] = nanmax(m[u], part[(size_t)(b + 16 * u) * 6 * W + t]);
    }
    for (; b < nblk; b += 16) m[0] = nanmax(m[0], part[(size_t)b * 6 * W + t]);
  }
  sm[ty][tx] = nanmax(nanmax(m[0], m[1]), nanmax(m[2], m[3]));
  __syncthreads();
  if (ty == 0) {
    double v = sm[0][tx];
#pragma unroll
    for (int k = 1; k < 16; k++) v = nanmax(v, sm[k][tx]);
    if (t < 6 * W) {
      const int r = t / W, pose = t - r * W;
      rowmax[6 * pose + r] = (unsigned long long)__double_as_longlong(v);
    } else if (t < npad) {
      rowmax[t] = 0ull;
    }
  }
}

template <int FORM, bool REGS, bool MAXR = false>
__global__ __launch_bounds__(256) void k_feature_factors(const double *__restrict__ cl,
                                                         const double *__restrict__ poses,
                                                         const double *__restrict__ feat, int W, int Wc, int npad, int f0,
                                                         int f1, double *__restrict__ Gt,
                                                         double *__restrict__ dpart, const int *__restrict__ slot, int staged,
                                                         double *__restrict__ rowmax_part = nullptr) {
  static_assert(!MAXR || REGS, "the row maxima live in the lane that owns the pose");
  double rm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, bad = 0.0;
  // staged (round 4, the default wherever the LDS has room): a lane's six values of a Gt column are 48 contiguous bytes, a wavefront's 64 poses
  // 3 KB -- written lane by lane as three 16-byte stores, every store instruction touches a THIRD of each 48-byte segment of 24 cache lines,
  // three times over.  Through a 3 KB staging block per wavefront in LDS the same bytes leave as three stores of 1 KB each, consecutive lanes
  // consecutive 16 bytes: 0.554 -> 0.444 ms at config 2 (2.24 GB at 5.05 TB/s, 0.80 of the copy rate; profiles/r04l_factors_staged.txt).
  // Bit for bit the same Gt.  (Contiguous feature ranges per workgroup instead of every gridDim-th feature: +0.02 ms, rejected.)
  constexpr int DACC = FORM == 0 ? DACC_LEFT : DACC_RIGHT;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  // blockIdx.y = chunk of Wc poses (one chunk = the whole window up to MAX_W_LDS poses)
  const int p0 = blockIdx.y * Wc, wc = min(Wc, W - p0);
  double *sp = sm;                 // [12][Wc] poses of the chunk
  double *sacc = sm + 12 * Wc;     // [DACC][Wc]
  double preg[12], racc[DACC];     // (REGS: the lane's own pose and accumulators)
  if (REGS) {
    const double *q = poses + 12 * (p0 + (threadIdx.x < (unsigned)wc ? (int)threadIdx.x : 0));
#pragma unroll
    for (int c = 0; c < 12; c++) preg[c] = q[c];
#pragma unroll
    for (int k = 0; k < DACC; k++) racc[k] = 0.0;
  }
  // the pose table, ten loads per lane in flight at a time (rolled, this copy was one memory round trip per iteration -- ~10 dependent
  // trips at W = 200 before a workgroup's first feature: tools/find_rolled_copies.py)
  if (!REGS)
  for (int t0 = 0; t0 < 12 * wc; t0 += 10 * (int)blockDim.x) {
    double pv[10];
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int t = t0 + j * (int)blockDim.x + (int)threadIdx.x;
      pv[j] = t < 12 * wc ? poses[12 * p0 + t] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 10; j++) {
      const int t = t0 + j * (int)blockDim.x + (int)threadIdx.x;
      if (t < 12 * wc) { const int il = t / 12, c = t - 12 * il; sp[c * Wc + il] = pv[j]; }
    }
  }
  if (!REGS) {
    for (int t = threadIdx.x; t < DACC * Wc; t += blockDim.x) sacc[t] = 0.0;
    __syncthreads();
  }

  // cluster of (feature a, pose i): ten coalesced streams, loaded one feature ahead of its use.  (Two features ahead -- two
  // register sets, the loop unrolled by two -- measured SLOWER at config 2: 0.584 vs 0.551 ms in round 3, and again on top of the
  // coalesced stores in round 4: 0.447 vs 0.437 ms, profiles/r04l_factors_staged.txt.  The kernel is not short of loads in flight.)
  double nxt[10];
  auto fetch = [&](int a, int i) {
    const double *ca = cl + (size_t)a * 10 * W + i;
#pragma unroll
    for (int c = 0; c < 10; c++) nxt[c] = ca[(size_t)c * W];
  };
  const int i_first = p0 + (threadIdx.x < (unsigned)wc ? (int)threadIdx.x : 0);
  const int a_begin = f0 + (int)blockIdx.x, a_end = f1, a_step = (int)gridDim.x;
  const int lane = threadIdx.x & 63;
  double *stg = sm + (12 + DACC) * Wc + (threadIdx.x >> 6) * 384;      // (staged only: 3 KB per wavefront behind the accumulators)
  if (a_begin < a_end) fetch(a_begin, i_first);

  for (int a = a_begin; a < a_end; a += a_step) {
    const double *f = feat + (size_t)a * FEAT_STRIDE;
    const FeatRec fr = {f[FT_NN], 1.0 / f[FT_NN], {f[FT_VBAR], f[FT_VBAR + 1], f[FT_VBAR + 2]}, {f[FT_U0], f[FT_U0 + 1], f[FT_U0 + 2]},
                        {f[FT_U1], f[FT_U1 + 1], f[FT_U1 + 2]}, {f[FT_U2], f[FT_U2 + 1], f[FT_U2 + 2]}, f[FT_C0], f[FT_C1], f[FT_C2], f[FT_COE]};
    const double *ca = cl + (size_t)a * 10 * W;
    double *g0 = Gt + (size_t)(3 * (slot ? slot[a] : a - f0)) * npad;      // slot: the block-sparse plan's column order

    for (int ilb = 0; ilb < wc; ilb += blockDim.x) {
      const int il = ilb + (int)threadIdx.x;
      if (il - lane >= wc) break;            // (a whole wavefront beyond the window)
      const bool act = il < wc;
      const int i = p0 + (act ? il : wc - 1);
      double col0[6], col1[6], col2[6];
      double P[6], v[3];
      if (!act) {
#pragma unroll
        for (int c = 0; c < 6; c++) P[c] = 0.0;
        v[0] = v[1] = v[2] = 0.0;
      } else if (il == (int)threadIdx.x) {         // first pose slot of this lane: prefetched
#pragma unroll
        for (int c = 0; c < 6; c++) P[c] = nxt[c];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = nxt[6 + c];
      } else {                               // W > blockDim.x: further slots are loaded in place
#pragma unroll
        for (int c = 0; c < 6; c++) P[c] = ca[(size_t)c * W + i];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = ca[(size_t)(6 + c) * W + i];
      }
      const double N = !act ? 0.0 : (il == (int)threadIdx.x ? nxt[9] : ca[(size_t)9 * W + i]);
      if (il == (int)threadIdx.x && a + a_step < a_end) fetch(a + a_step, i_first);
      if (REGS) obs_factors<FORM>(fr, P, v, N, preg, racc, 1, 0, col0, col1, col2);
      else obs_factors<FORM>(fr, P, v, N, sp, sacc, Wc, act ? il : wc - 1, col0, col1, col2);
      if (MAXR) track_rowmax(rm, bad, col0, col1, col2);
      // (Measured and rejected on the lane-by-lane stores, round 4, profiles/r04d_factors_ab.txt: streaming (nontemporal) stores -- 0.555 vs
      // 0.553 ms; the lane's pose in twelve registers instead of the LDS table, i.e. THREE workgroups per CU -- 0.565 vs 0.554.)
      if (!staged) {
        if (act) {
          store6(g0 + 6 * i, col0);
          store6(g0 + (size_t)npad + 6 * i, col1);
          store6(g0 + (size_t)2 * npad + 6 * i, col2);
        }
      } else {
        // the wavefront's block of a column: 6 x (its poses inside the window) doubles, contiguous from pose p0 + il - lane on
        const int nval = 6 * min(64, wc - (il - lane));
        double *gw = g0 + 6 * (size_t)(p0 + il - lane);
        auto flush = [&](const double col[6], double *gcol) {
          d2 *q = reinterpret_cast<d2 *>(stg + 6 * lane);
          d2 t0 = {col[0], col[1]}, t1 = {col[2], col[3]}, t2 = {col[4], col[5]};
          q[0] = t0; q[1] = t1; q[2] = t2;
          asm volatile("" ::: "memory");       // (one wavefront: its LDS operations are performed in order)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int idx = 2 * (64 * j + lane);
            const d2 w = *reinterpret_cast<const d2 *>(stg + idx);
            if (idx < nval) *reinterpret_cast<d2 *>(gcol + idx) = w;
          }
          asm volatile("" ::: "memory");
        };
        flush(col0, gw);
        flush(col1, gw + (size_t)npad);
        flush(col2, gw + (size_t)2 * npad);
      }
    }
  }
  double *dp = dpart + (size_t)blockIdx.x * DACC * W + p0;
  if (REGS) {                                        // a lane writes its own column of the partial sums: consecutive lanes, consecutive addresses
    if (threadIdx.x < (unsigned)wc) {
#pragma unroll
      for (int k = 0; k < DACC; k++) dp[(size_t)k * W + threadIdx.x] = racc[k];
      if (MAXR) publish_rowmax(rowmax_part, W, p0 + (int)threadIdx.x, rm, bad);
    }
    return;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < DACC * wc; t += blockDim.x) {
    const int k = t / wc, il = t - k * wc;
    dp[(size_t)k * W + il] = sacc[k * Wc + il];
  }
}

// ------------------------------------------------------------------------------------------------
// K1 + K1b + K2 in ONE pass over the clusters (round 3): the LM loop evaluates the residual at the trial poses (K1, K1b:
// 800 MB read at config 3) and, if the step is accepted, the factors at the very same poses one iteration later (K2: the same
// 800 MB again).  Here the trial evaluation also leaves the factors: a workgroup takes one feature at a time like K2 (a lane
// = a pose), but first reduces the lanes' world-frame clusters to the feature's moments, a dedicated wavefront turns them
// into the eigen record (the same Jacobi as K1b; ~4 us of dependent arithmetic -- it works on feature a + G while the pose
// lanes run the factors of feature a, whose record it produced one step earlier), and the pose lanes go on to the factors
// with the cluster still in their registers.  If the step is accepted the next Hessian evaluation starts at the SYRK; if
// not, the Gt of the current poses is still there (two Gt buffers).  Windows up to 256 poses (one pose per lane).
// ------------------------------------------------------------------------------------------------
template <int FORM, bool MAXR = false>
__global__ __launch_bounds__(320, 3) void k_moments_factors(const double *__restrict__ cl, const double *__restrict__ poses,
                                                         const double *__restrict__ fix, const double *__restrict__ coe_in, int W, int npad,
                                                         int F, double *__restrict__ Gt, double *__restrict__ dpart,
                                                         const int *__restrict__ slot, double *__restrict__ feat_out,
                                                         double *__restrict__ rpart, double *__restrict__ rowmax_part = nullptr) {
  constexpr int DACC = FORM == 0 ? DACC_LEFT : DACC_RIGHT;
  double rm[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, bad = 0.0;      // (MAXR: k_feature_factors)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int npl = (int)blockDim.x - 64, npw = npl >> 6;      // pose lanes / waves; the last wave is the eigen wave
  double *sp = sm;                                           // [12][W] poses
  double *sacc = sm + 12 * W;                                // [DACC][W]
  double *mom = sacc + DACC * W;                             // [2][4][10] partial moments per pose wave
  double *eig = mom + 80;                                    // [2][FEAT_STRIDE] eigen records
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool is_pose = tid < npl;
  for (int t = tid; t < 12 * W; t += blockDim.x) {
    const int i = t / 12, c = t - 12 * i;
    sp[c * W + i] = poses[t];
  }
  for (int t = tid; t < DACC * W; t += blockDim.x) sacc[t] = 0.0;
  __syncthreads();
  const int il = tid, G = gridDim.x, a0 = blockIdx.x;
  const bool has_pose = is_pose && il < W;
  double cur[10], nxt[10];
#pragma unroll
  for (int c = 0; c < 10; c++) cur[c] = nxt[c] = 0.0;
  auto fetch = [&](int a, double (&dst)[10]) {
    const double *ca = cl + (size_t)a * 10 * W + il;
#pragma unroll
    for (int c = 0; c < 10; c++) dst[c] = ca[(size_t)c * W];
  };
  // the lane's cluster in the world frame, summed over the wave -> mom[buf][wave][10]
  auto partial_moments = [&](const double (&c10)[10], int buf) {
    double m[10];
#pragma unroll
    for (int c = 0; c < 10; c++) m[c] = 0.0;
    if (has_pose && (int)c10[9] > 0) {
      double R[9], p[3];
#pragma unroll
      for (int c = 0; c < 9; c++) R[c] = sp[c * W + il];
#pragma unroll
      for (int c = 0; c < 3; c++) p[c] = sp[(9 + c) * W + il];
      Obs o;
      to_world(c10, c10 + 6, c10[9], R, p, o);
#pragma unroll
      for (int c = 0; c < 6; c++) m[c] = o.Pw[c];
#pragma unroll
      for (int c = 0; c < 3; c++) m[6 + c] = o.b[c];
      m[9] = c10[9];
    }
#pragma unroll
    for (int c = 0; c < 10; c++) m[c] = wave_sum(m[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 10; c++) mom[(buf * 4 + wv) * 10 + c] = m[c];
    }
  };
  double res = 0.0;
  // eigen record of feature a from the partial moments (K1b's arithmetic), wave-uniform
  auto eigen_step = [&](int a, int buf) {
    double c[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
      double t = mom[(buf * 4) * 10 + k];
      for (int w = 1; w < npw; w++) t += mom[(buf * 4 + w) * 10 + k];
      c[k] = t;
    }
    if (fix) {
#pragma unroll
      for (int k = 0; k < 10; k++) c[k] += fix[(size_t)a * 10 + k];
    }
    const double NN = c[9], inv = 1.0 / NN;
    const double vb0 = c[6] * inv, vb1 = c[7] * inv, vb2 = c[8] * inv;
    double lam[3], U[3][3];
    eig3_jacobi(c[0] * inv - vb0 * vb0, c[1] * inv - vb0 * vb1, c[2] * inv - vb0 * vb2, c[3] * inv - vb1 * vb1,
                c[4] * inv - vb1 * vb2, c[5] * inv - vb2 * vb2, lam, U);
    const double w = coe_in[a];
    res += w * lam[0];
    double rec[FEAT_STRIDE];
#pragma unroll
    for (int k = 0; k < FEAT_STRIDE; k++) rec[k] = 0.0;
    rec[FT_NN] = NN;
    rec[FT_VBAR] = vb0; rec[FT_VBAR + 1] = vb1; rec[FT_VBAR + 2] = vb2;
    rec[FT_LAM] = lam[0]; rec[FT_LAM + 1] = lam[1]; rec[FT_LAM + 2] = lam[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { rec[FT_U0 + k] = U[k][0]; rec[FT_U1 + k] = U[k][1]; rec[FT_U2 + k] = U[k][2]; }
    rec[FT_C0] = sqrt(2.0 * w) * inv;
    rec[FT_C1] = sqrt(2.0 * w / (lam[1] - lam[0])) * inv;
    rec[FT_C2] = sqrt(2.0 * w / (lam[2] - lam[0])) * inv;
    rec[FT_COE] = w;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < FEAT_STRIDE; k++) { eig[buf * FEAT_STRIDE + k] = rec[k]; feat_out[(size_t)a * FEAT_STRIDE + k] = rec[k]; }
    }
  };

  if (a0 < F) {
    if (has_pose) fetch(a0, cur);
    if (is_pose) partial_moments(cur, 0);
    if (has_pose && a0 + G < F) fetch(a0 + G, nxt);
  }
  __syncthreads();
  if (!is_pose && a0 < F) eigen_step(a0, 0);
  __syncthreads();
  int b = 0;
  for (int a = a0; a < F; a += G, b ^= 1) {
    const bool has_next = a + G < F;
    if (is_pose && has_next) partial_moments(nxt, b ^ 1);
    __syncthreads();                                           // the next feature's partial moments are in LDS
    if (!is_pose) {
      if (has_next) eigen_step(a + G, b ^ 1);
    } else {
      const double *f = eig + b * FEAT_STRIDE;
      const FeatRec fr = {f[FT_NN], 1.0 / f[FT_NN], {f[FT_VBAR], f[FT_VBAR + 1], f[FT_VBAR + 2]}, {f[FT_U0], f[FT_U0 + 1], f[FT_U0 + 2]},
                          {f[FT_U1], f[FT_U1 + 1], f[FT_U1 + 2]}, {f[FT_U2], f[FT_U2 + 1], f[FT_U2 + 2]}, f[FT_C0], f[FT_C1], f[FT_C2], f[FT_COE]};
      if (has_pose) {
        double col0[6], col1[6], col2[6];
        obs_factors<FORM>(fr, cur, cur + 6, cur[9], sp, sacc, W, il, col0, col1, col2);
        if (MAXR) track_rowmax(rm, bad, col0, col1, col2);
        double *g0 = Gt + (size_t)(3 * (slot ? slot[a] : a)) * npad;
        store6(g0 + 6 * il, col0);
        store6(g0 + (size_t)npad + 6 * il, col1);
        store6(g0 + (size_t)2 * npad + 6 * il, col2);
#pragma unroll
        for (int c = 0; c < 10; c++) cur[c] = nxt[c];
        if (a + 2 * G < F) fetch(a + 2 * G, nxt);
      }
    }
    __syncthreads();                                           // the next record is in LDS; this one and the partial moments are free
  }
  double *dp = dpart + (size_t)blockIdx.x * DACC * W;
  for (int t = tid; t < DACC * W; t += blockDim.x) dp[t] = sacc[t];
  if (MAXR && has_pose) publish_rowmax(rowmax_part, W, il, rm, bad);
  if (!is_pose && lane == 0) rpart[blockIdx.x] = res;
}

// poses per workgroup of the factor kernel: the whole window while its accumulators fit in LDS, else chunks
int factors_chunk(int W) { return W <= MAX_W_LDS ? W : 256; }

static size_t factors_lds(int W, int form) {
  const int dacc = form == 0 ? DACC_LEFT : DACC_RIGHT;
  return (size_t)(12 + dacc) * factors_chunk(W) * sizeof(double);
}
constexpr size_t FACTORS_STAGE_BYTES = 4 * 384 * sizeof(double);      // a 3 KB staging block per wavefront
// Coalesced Gt stores through LDS where the kernel is bandwidth-bound -- a million observations or more (config 2: ten million) -- and the
// staging blocks fit beside the accumulators (not for 470 < W <= 480).  Below that a workgroup sees a handful of features and the extra LDS round
// trip per feature shows instead: the shipped window (0.4 M observations) 0.029 -> 0.037 ms (profiles/r04m_realshape_step.txt).
// BALM_FACTORS_STAGE=0 / 1 forces either (A/B runs, tests).
static bool factors_staged(int W, int nfeat, int form) {
  if (factors_lds(W, form) + FACTORS_STAGE_BYTES > 160 * 1024) return false;
  if (const char *es = getenv("BALM_FACTORS_STAGE")) return es[0] != '0';
  return (long)nfeat * W >= 1000000;
}

int factors_grid(int W, int nfeat, int form) {
  size_t lds = factors_lds(W, form) + (factors_staged(W, nfeat, form) ? FACTORS_STAGE_BYTES : 0);
  int per_cu = (int)(160 * 1024 / lds);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  int grid = 256 * per_cu;
  // (grids of 256 / 768 / 1024 workgroups instead of 256 per resident workgroup of a CU: 0.72 / 0.65 / 0.55 vs 0.54 ms, profiles/r04d_factors_ab.txt)
  if (grid > nfeat) grid = nfeat;
  if (grid < 1) grid = 1;
  return grid;
}

// Dynamic LDS above the default 64 KiB (k_world_moments: windows above ~680 poses; k_feature_factors: above ~200).
// The attribute belongs to the CURRENT device: balm_create calls this once per context, after hipSetDevice.
hipError_t prepare_device_accum() {
  hipError_t e = hipFuncSetAttribute((const void *)k_world_moments, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_feature_factors<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_feature_factors<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_feature_factors<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_feature_factors<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_moments_factors<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_moments_factors<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_moments_factors<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_moments_factors<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}

// the fused trial evaluation: residual partials (one per workgroup: returns their number), eigen records, Gt, per-pose partials
int launch_moments_factors(hipStream_t s, int form, const double *cl, const double *poses, const double *fix, const double *coe, int W,
                           int npad, int F, double *Gt, double *dpart, int nblk, const int *slot, double *feat, double *rpart,
                           unsigned long long *rowmax, double *rowmax_part) {
  const int dacc = form == 0 ? DACC_LEFT : DACC_RIGHT;
  const size_t lds = (size_t)((12 + dacc) * W + 80 + 2 * FEAT_STRIDE) * sizeof(double);
  const int bs = (W <= 64 ? 64 : (W <= 128 ? 128 : 256)) + 64;
  if (rowmax) {          // (the INT8 product's row maxima ride along: per workgroup, then one small reduction)
    if (form == 0)
      hipLaunchKernelGGL((k_moments_factors<0, true>), dim3(nblk), dim3(bs), lds, s, cl, poses, fix, coe, W, npad, F, Gt, dpart, slot, feat, rpart, rowmax_part);
    else
      hipLaunchKernelGGL((k_moments_factors<1, true>), dim3(nblk), dim3(bs), lds, s, cl, poses, fix, coe, W, npad, F, Gt, dpart, slot, feat, rpart, rowmax_part);
    hipLaunchKernelGGL(k_rowmax_reduce, dim3((npad + 63) / 64), dim3(1024), 0, s, rowmax_part, nblk, W, npad, rowmax);
    return nblk;
  }
  if (form == 0)
    hipLaunchKernelGGL((k_moments_factors<0, false>), dim3(nblk), dim3(bs), lds, s, cl, poses, fix, coe, W, npad, F, Gt, dpart, slot, feat, rpart, nullptr);
  else
    hipLaunchKernelGGL((k_moments_factors<1, false>), dim3(nblk), dim3(bs), lds, s, cl, poses, fix, coe, W, npad, F, Gt, dpart, slot, feat, rpart, nullptr);
  return nblk;
}

// rowmax (may be null): where the kernel variant that tracks them exists -- the left form with a pose per lane -- the rows' largest |entries|
// are left there (npad entries; rowmax_part: 6 W doubles per workgroup of scratch) and the call returns true
bool launch_factors(hipStream_t s, int form, const double *cl, const double *poses, const double *feat, int W,
                    int npad, int f0, int f1, double *Gt, double *dpart, int nblk, const int *slot, unsigned long long *rowmax, double *rowmax_part) {
  const int Wc = factors_chunk(W), chunks = (W + Wc - 1) / Wc;
  size_t lds = factors_lds(W, form);
  int bs = W <= 64 ? 64 : (W <= 128 ? 128 : 256);
  const int staged = factors_staged(W, f1 - f0, form);
  if (staged) lds += FACTORS_STAGE_BYTES;
  const char *er = getenv("BALM_FACTORS_REGS");                     // 0: the accumulators in LDS at every window (A/B, tests)
  const bool regs = form == 0 && chunks == 1 && W <= bs && !(er && er[0] == '0');
  if (rowmax && regs) {
    hipLaunchKernelGGL((k_feature_factors<0, true, true>), dim3(nblk, chunks), dim3(bs), lds, s, cl, poses, feat, W, Wc, npad, f0, f1, Gt, dpart, slot, staged, rowmax_part);
    hipLaunchKernelGGL(k_rowmax_reduce, dim3((npad + 63) / 64), dim3(1024), 0, s, rowmax_part, nblk, W, npad, rowmax);
    return true;
  }
#define BALM_FACTORS(F, R) hipLaunchKernelGGL((k_feature_factors<F, R, false>), dim3(nblk, chunks), dim3(bs), lds, s, cl, poses, feat, W, Wc, npad, f0, f1, Gt, dpart, slot, staged, nullptr)
  if (form == 0) { if (regs) BALM_FACTORS(0, true); else BALM_FACTORS(0, false); }
  else BALM_FACTORS(1, false);
#undef BALM_FACTORS
  return false;
}

// ------------------------------------------------------------------------------------------------
// K3: hessian_syrk.  part[sg][job] = Gt[I-rows, kslice] * Gt[J-rows, kslice]^T for the off-diagonal
// 80x80 tiles of the upper triangle, and the upper 16x16 sub-tiles of the diagonal blocks packed 25 to a
// job.  One wavefront owns 25 v_mfma_f64_16x16x4_f64 accumulator tiles (200 AGPRs, one wave per SIMD) over
// one k-slice; no LDS, no barriers.
// Operands come straight from L2: lane l of an MFMA operand is Gt[k0 + (l>>4)][row0 + (l&15)] --
// four 128-byte row segments per instruction.
// ------------------------------------------------------------------------------------------------
#include "syrk_mfma_asm.inc"

__device__ __forceinline__ void load5(const double *__restrict__ p, double (&x)[TM]) {
#pragma unroll
  for (int r = 0; r < TM; r++) x[r] = p[16 * r];
}

// The 200 accumulator registers are pinned to AGPRs a0..a199 by generated inline asm
// (gen/gen_syrk_asm.py explains why); the compiler only sees the operand loads.  Operands are
// prefetched NBUF-1 k-steps ahead (a k-step = 25 MFMAs = 1600 issue cycles) through a register ring.
constexpr int SYRK_NBUF = 4;

// Diagonal tiles run the same 25-MFMA sweep as the others (their lower half is computed and ignored): a
// 15-MFMA triangular sweep finishes ~1.7x earlier, the wave leaves the lockstep of its k-slice and both it and
// its successor fetch their rows alone -- measured 4.15 GB vs 2.63 GB fetched per launch and 2.3 % slower.
__device__ __forceinline__ void syrk_sweep(const double *__restrict__ pa, const double *__restrict__ pb, size_t step,
                                           int nsteps) {
  double a[SYRK_NBUF][TM], b[SYRK_NBUF][TM];
#pragma unroll
  for (int i = 0; i < SYRK_NBUF - 1; i++) {
    load5(pa + i * step, a[i]);
    load5(pb + i * step, b[i]);
  }
  pa += (SYRK_NBUF - 1) * step;
  pb += (SYRK_NBUF - 1) * step;
  for (int s = 0; s < nsteps; s += SYRK_NBUF) {       // nsteps is a multiple of SYRK_NBUF
#pragma unroll
    for (int j = 0; j < SYRK_NBUF; j++) {
      const int nb = (j + SYRK_NBUF - 1) % SYRK_NBUF;
      load5(pa, a[nb]);                               // the last steps prefetch past the slice (allocated)
      load5(pb, b[nb]);
      pa += step; pb += step;
      BALM_SYRK_MFMA_FULL(a[j], b[j])
    }
  }
}

// A "mixed" wave: 25 upper sub-tile pairs of up to three consecutive diagonal blocks (syrk_mfma_asm.inc, gen/).
#define BALM_DEFINE_MIXED_SWEEP(V)                                                                                  \
  __device__ __forceinline__ void syrk_sweep_mixed##V(const double *__restrict__ p, size_t step, int nsteps) {    \
    double x[SYRK_NBUF][BALM_SYRK_MIX##V##_NLOAD];                                                                  \
    _Pragma("unroll") for (int i = 0; i < SYRK_NBUF - 1; i++) {                                                    \
      const double *q = p + i * step;                                                                               \
      BALM_SYRK_MIX##V##_LOAD(q, x[i])                                                                              \
    }                                                                                                               \
    p += (SYRK_NBUF - 1) * step;                                                                                    \
    for (int s = 0; s < nsteps; s += SYRK_NBUF) {                                                                   \
      _Pragma("unroll") for (int j = 0; j < SYRK_NBUF; j++) {                                                      \
        const int nb = (j + SYRK_NBUF - 1) % SYRK_NBUF;                                                             \
        BALM_SYRK_MIX##V##_LOAD(p, x[nb])                                                                           \
        p += step;                                                                                                  \
        BALM_SYRK_MIX##V##_MFMA(x[j])                                                                               \
      }                                                                                                             \
    }                                                                                                               \
  }
BALM_DEFINE_MIXED_SWEEP(1)
BALM_DEFINE_MIXED_SWEEP(2)
BALM_DEFINE_MIXED_SWEEP(3)
#undef BALM_DEFINE_MIXED_SWEEP

// One wavefront = one job (an off-diagonal 80x80 tile, or 25 upper sub-tiles of the diagonal blocks: jobs[4 j] = type,
// block I or base block, block J).  syrk_accumulate adds nsteps k-steps from column k_begin on to the wave's pinned
// accumulators; syrk_write_out stores the 80x80 partial tile.
__device__ __forceinline__ void syrk_accumulate(const double *__restrict__ Gt, int npad, int type, int I, int J,
                                                size_t k_begin, int nsteps) {
  const int lane = threadIdx.x;
  const double *base = Gt + (k_begin + (lane >> 4)) * (size_t)npad + (lane & 15);
  const double *pa = base + I * TILE;
  const double *pb = base + J * TILE;
  const size_t step = (size_t)4 * npad;
  if (type == 0) syrk_sweep(pa, pb, step, nsteps);
  else if (type == 1) syrk_sweep_mixed1(pa, step, nsteps);
  else if (type == 2) syrk_sweep_mixed2(pa, step, nsteps);
  else syrk_sweep_mixed3(pa, step, nsteps);
}

__device__ __forceinline__ void syrk_write_out(double *__restrict__ out) {
  // MFMA (16 passes) -> v_accvgpr_read needs wait states the assembler will not insert for asm
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  out += threadIdx.x;                                                        // register-major: coalesced
#define BALM_X(T)                                                                             \
  {                                                                                           \
    unsigned U[8];                                                                            \
    BALM_SYRK_READ_##T(U);                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; e++) out[(T * 4 + e) * 64] = __hiloint2double(U[2 * e + 1], U[2 * e]); \
  }
  BALM_SYRK_FOR_TILES(BALM_X)
#undef BALM_X
}

// XCD-aware remap: hardware places workgroup b on XCD b % 8; every XCD gets a contiguous run of logical workgroups.
__device__ __forceinline__ long xcd_remap(long bid, long nblocks) {
  const long q = nblocks >> 3, r = nblocks & 7, x = bid & 7;       // XCD x runs q (+1 if x < r) logical workgroups
  return x * q + (x < r ? x : r) + (bid >> 3);
}

__global__ __launch_bounds__(64) void k_hessian_syrk(const double *__restrict__ Gt, int npad, int njobs,
                                                     const int *__restrict__ jobs, int nsteps, long nblocks,
                                                     double *__restrict__ part) {
  // Dense plan: one wavefront per (job, k-slice).  The remap gives an XCD ALL jobs of one k-slice after the other:
  // an XCD holds 128 of these one-wave workgroups (4 per CU), i.e. the jobs of a slice run side by side and sweep
  // k in lockstep (they are all MFMA-paced, 25 MFMAs per k-step each), so each 128-byte line of Gt is pulled into
  // that XCD's L2 once and serves the ~15 jobs that need it.
  const long bid = xcd_remap(blockIdx.x, nblocks);
  const int tile = (int)(bid % njobs);
  const int sg = (int)(bid / njobs);
  BALM_SYRK_ZERO_ACC();
  syrk_accumulate(Gt, npad, jobs[4 * tile], jobs[4 * tile + 1], jobs[4 * tile + 2], (size_t)sg * nsteps * 4, nsteps);
  syrk_write_out(part + ((size_t)sg * njobs + tile) * TILE_ELEMS);
}

// Block-sparse plan (real co-visibility: the reference only visits observed pose pairs, bavoxel.hpp:365,404-418 `N != 0`
// guards).  The columns of Gt are ordered so that features observing the same stretch of the trajectory are neighbours
// and cut into chunks of nsteps k-steps; a job only needs the chunks whose features touch both of its row blocks.  An
// item = (job, a run of that job's chunk list): the wave walks its chunks and keeps accumulating in the same registers,
// so the number of partial tiles (51 KB each, written here and read by the reduction) does not grow with the sparsity
// granularity.  items[4 i] = job, first entry of chunk_ids, entry count, output slot (slots are grouped by job).
__global__ __launch_bounds__(64) void k_hessian_syrk_sparse(const double *__restrict__ Gt, int npad,
                                                            const int *__restrict__ jobs, const int *__restrict__ items,
                                                            const int *__restrict__ chunk_ids, int nsteps, long nitems,
                                                            double *__restrict__ part) {
  const long bid = xcd_remap(blockIdx.x, nitems);
  const int tile = items[4 * bid], first = items[4 * bid + 1], count = items[4 * bid + 2], slot = items[4 * bid + 3];
  const int type = jobs[4 * tile], I = jobs[4 * tile + 1], J = jobs[4 * tile + 2];
  BALM_SYRK_ZERO_ACC();
  for (int c = 0; c < count; c++)
    syrk_accumulate(Gt, npad, type, I, J, (size_t)chunk_ids[first + c] * nsteps * 4, nsteps);
  syrk_write_out(part + (size_t)slot * TILE_ELEMS);
}

SyrkPlan plan_syrk(int ntiles, long K) {
  SyrkPlan p;
  long steps = (K + 3) / 4;                         // MFMA k-steps (4 columns of Gt each)
  if (steps < 1) steps = 1;
  long max_sg = steps / 64;                         // keep >= 256 columns per wave ...
  if (max_sg < 1) max_sg = 1;
  {
    // ... unless that leaves the chip mostly EMPTY: a small window has few jobs per k-slice (three at 20 poses, 13 at 64), and with >= 64
    // k-steps per wave a 20-pose window with 150 features ran its whole K range in ONE wave per job -- 113 k-steps x 25 MFMAs in series,
    // 83 of the iteration's 155 us (profiles/r04w_small_lm_kernels.txt).  While one round of the 1024 wave slots is not full: the shortest
    // waves (whole turns of the prefetch ring) whose slices still fit that round, and as many slices as that length needs -- so the padding
    // of K stays below one wave's length.  Their partial tiles (51 KB each, <= 52 MB in all) are noise next to the serial MFMAs they replace.
    // (A/B against >= 64 k-steps per wave: W = 20 / F = 150 0.156 -> 0.081 ms per iteration, profiles/r04x_small_windows_syrk_plan.txt)
    const long one_round = 1024 / ntiles > 1 ? 1024 / ntiles : 1;
    if (max_sg < one_round) {
      long nst = SYRK_NBUF;
      while ((steps + nst - 1) / nst > one_round) nst += SYRK_NBUF;
      p.SG = (int)((steps + nst - 1) / nst);
      p.nsteps = (int)nst;
      p.Kpad = (int)(p.SG * nst * 4);
      p.nblocks = (long)p.SG * ntiles;
      return p;
    }
  }
  // ~4 resident rounds of the 1024 wave slots (128 per XCD): pick the slice count in that
  // neighbourhood whose last round is fullest (measured with 114 jobs per slice: 35 / 44 / 53 slices = 4 / 5 / 6
  // rounds are within 0.7 % of each other, the extra partial tiles of the larger counts cost it back in the reduce)
  long base = (4096 + ntiles - 1) / ntiles, sg = base;
  double best = -1.0;
  for (long c = (base > 6 ? base - 6 : 1); c <= base + 6; c++) {
    const double per_xcd = (double)ntiles * c / 8.0;
    const double rounds = (double)(long)((per_xcd + 127.0) / 128.0);
    const double eff = per_xcd / (rounds * 128.0);
    if (eff > best + 1e-9) { best = eff; sg = c; }
  }
  if (sg > max_sg) sg = max_sg;
  // small problems (the shipped window: 1 700 k-steps): four rounds of short waves would mostly write and re-read partial
  // tiles (51 KB each); one round of longer waves does the same MFMAs with a quarter of that traffic
  if (steps / sg < 128 && sg > 1) {
    long one_round = 1024 / ntiles;
    if (one_round < 1) one_round = 1;
    if (one_round < sg) sg = one_round;
  }
  long nst = (steps + sg - 1) / sg;
  nst = (nst + SYRK_NBUF - 1) / SYRK_NBUF * SYRK_NBUF;      // whole turns of the prefetch ring
  p.SG = (int)sg;
  p.nsteps = (int)nst;                     // k-steps per wave
  p.Kpad = (int)(sg * nst * 4);
  p.nblocks = sg * ntiles;
  return p;
}

void launch_syrk_sparse(hipStream_t s, const double *Gt, int npad, const int *jobs, const int *items, const int *chunk_ids,
                        int nsteps, long nitems, double *part) {
  hipLaunchKernelGGL(k_hessian_syrk_sparse, dim3((unsigned)nitems), dim3(64), 0, s, Gt, npad, jobs, items, chunk_ids, nsteps,
                     nitems, part);
}

void launch_syrk(hipStream_t s, const double *Gt, int npad, int ntiles, const int *tileIJ, const SyrkPlan &p,
                 double *part) {
  // (Measured and rejected, round 4, profiles/r04f_overlap_ab.txt: the launch cut at its rounds of 1024 workgroups with the factor
  // kernel's later slabs on a second stream beside them -- factors + syrk 3.70 instead of 3.64 ms per step at config 2: the SYRK keeps
  // the FP64 datapath busy 97 % of the time, and the factor kernel's f64 VALU work shares that datapath: it is added, not hidden.)
  hipLaunchKernelGGL(k_hessian_syrk, dim3((unsigned)p.nblocks), dim3(64), 0, s, Gt, npad, ntiles, tileIJ,
                     p.nsteps, p.nblocks, part);
}

// ------------------------------------------------------------------------------------------------
// K4a: deterministic reductions into the all-reduce payload  red = [tiles | dacc | r]
// ------------------------------------------------------------------------------------------------
template <class Sink>
__device__ __forceinline__ void reduce_tiles(const double *__restrict__ part, int SG, long tile_total, Sink sink,
                                             int bid, int nb) {
  for (long t = (long)bid * blockDim.x + threadIdx.x; t < tile_total; t += (long)nb * blockDim.x) {
    // fixed summation order (deterministic), four independent chains to keep loads in flight
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double *pp = part + t;
    int g = 0;
    for (; g + 3 < SG; g += 4) {
      s0 += pp[(size_t)g * tile_total];
      s1 += pp[(size_t)(g + 1) * tile_total];
      s2 += pp[(size_t)(g + 2) * tile_total];
      s3 += pp[(size_t)(g + 3) * tile_total];
    }
    for (; g < SG; g++) s0 += pp[(size_t)g * tile_total];
    sink(t, (s0 + s1) + (s2 + s3));
  }
}

// The same sums with FOUR LANES per element, one per chain (round 4): on a small window plan_syrk cuts K into many short waves (up to 341 slices at
// 20 poses) over few tile elements (19 200), and one thread per element then walks its 341 partials four at a time -- ~85 dependent trips, the
// largest kernel of a 20-pose / 3 000-feature iteration beside the solve.  Lane c of a quad adds chain c (g = c, c + 4, ...; the tail on chain 0) with
// eight loads in flight, two xor-shuffles form (s0 + s1) + (s2 + s3): the additions of reduce_tiles in its order, bit for bit.
template <class Sink>
__device__ __forceinline__ void reduce_tiles_quads(const double *__restrict__ part, int SG, long tile_total, Sink sink, int bid, int nb) {
  const int c = threadIdx.x & 3;
  const int full = SG / 4;                           // rounds of four
  // (a quad = four consecutive lanes of one wavefront: they share t and leave the loop together, so the shuffles below stay inside live lanes)
  for (long t = ((long)bid * blockDim.x + threadIdx.x) >> 2; t < tile_total; t += ((long)nb * blockDim.x) >> 2) {
    const double *pp = part + t + (size_t)c * tile_total;
    double s = 0.0;
    int m = 0;
    for (; m + 3 < full; m += 4) {                    // four loads of the chain in flight (sixteen per element), added in order
      const double v0 = pp[(size_t)(4 * m) * tile_total], v1 = pp[(size_t)(4 * m + 4) * tile_total],
                   v2 = pp[(size_t)(4 * m + 8) * tile_total], v3 = pp[(size_t)(4 * m + 12) * tile_total];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; m < full; m++) s += pp[(size_t)(4 * m) * tile_total];
    if (c == 0)
      for (int g = 4 * full; g < SG; g++) s += part[(size_t)g * tile_total + t];
    s += __shfl_xor(s, 1, 64);                       // (s0 + s1), (s2 + s3)
    s += __shfl_xor(s, 2, 64);
    if (c == 0) sink(t, s);
  }
}

// block-sparse plan: the partial tiles of job j are slots ptr[j] .. ptr[j+1]-1 (chunk order: deterministic)
template <class Sink>
__device__ __forceinline__ void reduce_tiles_csr(const double *__restrict__ part, const int *__restrict__ ptr, long tile_total,
                                                 Sink sink, int bid, int nb) {
  for (long t = (long)bid * blockDim.x + threadIdx.x; t < tile_total; t += (long)nb * blockDim.x) {
    const int job = (int)(t / TILE_ELEMS);
    const long e = t - (long)job * TILE_ELEMS;
    const int s0 = ptr[job], s1 = ptr[job + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int g = s0;
    for (; g + 3 < s1; g += 4) {
      a0 += part[(size_t)g * TILE_ELEMS + e];
      a1 += part[(size_t)(g + 1) * TILE_ELEMS + e];
      a2 += part[(size_t)(g + 2) * TILE_ELEMS + e];
      a3 += part[(size_t)(g + 3) * TILE_ELEMS + e];
    }
    for (; g < s1; g++) a0 += part[(size_t)g * TILE_ELEMS + e];
    sink(t, (a0 + a1) + (a2 + a3));
  }
}

// per-pose accumulators: sum over the feature_factors workgroups.  64 outputs per workgroup, the
// workgroup index range split over the four waves, eight loads in flight per lane.
__device__ __forceinline__ void reduce_dacc(const double *__restrict__ dpart, int nblk, int dacc_len, int dacc_cap,
                                            const double *__restrict__ rpart, int nr, double *__restrict__ red_dacc,
                                            double *__restrict__ red_r, int bid) {
  __shared__ double sq[256];
  const int jl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int j = bid * 64 + jl;
  double s = 0.0;
  if (j < dacc_len) {
    const int chunk = (nblk + 3) / 4;
    const int b0 = q * chunk, b1 = min(nblk, b0 + chunk);
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = 0.0;
    int b = b0;
    for (; b + 7 < b1; b += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] += dpart[(size_t)(b + k) * dacc_len + j];
    }
    for (; b < b1; b++) a[0] += dpart[(size_t)b * dacc_len + j];
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  sq[threadIdx.x] = s;
  __syncthreads();
  if (q == 0 && j < dacc_len) red_dacc[j] = (sq[jl] + sq[64 + jl]) + (sq[128 + jl] + sq[192 + jl]);
  if (q == 0 && j >= dacc_len && j < dacc_cap) red_dacc[j] = 0.0;      // the payload's unused accumulator slots (left form: 27 of 30)
  if (bid == 0) {
    __syncthreads();
    double r = 0.0;
    for (int t = threadIdx.x; t < nr; t += 256) r += rpart[t];
    sq[threadIdx.x] = r;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if (threadIdx.x < k) sq[threadIdx.x] += sq[threadIdx.x + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) { red_r[0] = sq[0]; red_r[1] = 0.0; }
  }
}

// K4a in ONE launch (round 3: the two reductions are independent -- they used to be two launches one after the other): blocks
// [0, tile_blocks) sum the split-K partial tiles, the blocks behind them the per-pose accumulators and the residual partials
__global__ __launch_bounds__(256) void k_reduce_all(const double *__restrict__ part, int SG, const int *__restrict__ csr_ptr, long tile_total,
                                                    int tile_blocks, const double *__restrict__ dpart, int nblk, int dacc_len, int dacc_cap,
                                                    const double *__restrict__ rpart, int nr, double *__restrict__ red, long dacc_off, long r_off, int quads) {
  if ((int)blockIdx.x < tile_blocks) {
    auto to_payload = [&](long t, double v) { red[t] = v; };
    if (csr_ptr) reduce_tiles_csr(part, csr_ptr, tile_total, to_payload, blockIdx.x, tile_blocks);
    else if (quads) reduce_tiles_quads(part, SG, tile_total, to_payload, blockIdx.x, tile_blocks);
    else reduce_tiles(part, SG, tile_total, to_payload, blockIdx.x, tile_blocks);
  } else {
    reduce_dacc(dpart, nblk, dacc_len, dacc_cap, rpart, nr, red + dacc_off, red + r_off, (int)blockIdx.x - tile_blocks);
  }
}

void launch_reduce(hipStream_t s, const double *part, int SG, long tile_total, const double *dpart, int nblk,
                   int dacc_len, const double *rpart, int nr, double *red, long dacc_off, long r_off, const int *csr_ptr) {
  const int dacc_cap = (int)(r_off - dacc_off);        // every slot of the payload between the tiles and the residual is written
  // few tile elements, many k-slices (small windows since plan_syrk fills a round of wave slots with short waves): four lanes per element
  static const bool no_quads = getenv("BALM_REDUCE_QUADS") && getenv("BALM_REDUCE_QUADS")[0] == '0';      // A/B
  const int quads = !csr_ptr && !no_quads && tile_total <= 32768 && SG >= 8;      // (windows of up to ~30 poses: 0.121 -> 0.115 ms per iteration at W = 20 / F = 3000; at 64 poses one thread per element is the faster form: profiles/r04x_small_windows_syrk_plan.txt)
  int grid = (int)(((quads ? 4 : 1) * tile_total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_reduce_all, dim3(grid + (dacc_cap + 63) / 64), dim3(256), 0, s, part, SG, csr_ptr, tile_total, grid, dpart, nblk, dacc_len,
                     dacc_cap, rpart, nr, red, dacc_off, r_off, quads);
}

// ------------------------------------------------------------------------------------------------
// K4b: H = blockdiag(B_i) - (Gt Gt^T), mirrored to the lower triangle (bavoxel.hpp:422-424);
// g = per-pose gradient sums.  One lane per tile element.
// MFMA f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sym3(int r, int c) {   // index into xx xy xz yy yz zz
  if (r > c) { int t = r; r = c; c = t; }
  return r == 0 ? c : (r == 1 ? 2 + c : 5);
}

template <int FORM>
__device__ __forceinline__ double blockdiag_at(const double *__restrict__ dacc, int W, int i, int r, int c) {
  // left : 6 grad | TL sym(6) | TR (9, rows 0..2 x cols 3..5) | BR sym(6)
  // right: 6 grad | TL full(9) | TR (9) | BR sym(6)
  const double *d = dacc + i;
  if (FORM == 0) {
    if (r < 3 && c < 3) return d[(size_t)(6 + sym3(r, c)) * W];
    if (r < 3 && c >= 3) return d[(size_t)(12 + 3 * r + (c - 3)) * W];
    if (r >= 3 && c < 3) return d[(size_t)(12 + 3 * c + (r - 3)) * W];
    return d[(size_t)(21 + sym3(r - 3, c - 3)) * W];
  } else {
    if (r < 3 && c < 3) return d[(size_t)(6 + 3 * r + c) * W];
    if (r < 3 && c >= 3) return d[(size_t)(15 + 3 * r + (c - 3)) * W];
    if (r >= 3 && c < 3) return d[(size_t)(15 + 3 * c + (r - 3)) * W];
    return d[(size_t)(24 + sym3(r - 3, c - 3)) * W];
  }
}

template <int FORM>
__global__ __launch_bounds__(256) void k_assemble(const double *__restrict__ red, long dacc_off,
                                                  const int *__restrict__ tileIJ, int ntiles, int W,
                                                  double *__restrict__ H, double *__restrict__ g, const double *__restrict__ r_in,
                                                  double *__restrict__ r_out) {
  const int n = 6 * W;
  if (r_out && blockIdx.x == 0 && threadIdx.x == 0) *r_out = *r_in;      // the residual of the (summed) payload -> the LM loop's scalars
  const long total = (long)ntiles * TILE_ELEMS;
  const double *dacc = red + dacc_off;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int tile = (int)(t / TILE_ELEMS);
    const int e = (int)(t - (long)tile * TILE_ELEMS);
    const int lane = e & 63, slot = e >> 6;
    const int reg = slot & 3, mt = slot >> 2;
    const int rc = tileIJ[tile * 25 + mt];              // global 16-row sub-tile coordinates of accumulator tile mt
    const int row = (rc >> 16) * 16 + (lane >> 4) + 4 * reg;
    const int col = (rc & 0xffff) * 16 + (lane & 15);
    if (row >= n || col >= n || row > col) continue;   // diagonal sub-tiles: only r <= c is used
    double val = -red[t];
    const int pi = row / 6, pj = col / 6;
    if (pi == pj) {
      const int r = row - 6 * pi, c = col - 6 * pi;
      val += blockdiag_at<FORM>(dacc, W, pi, r, c);
      if (FORM == 1 && r != c) {
        // the right form's diagonal block is not forced symmetric in the reference (:129)
        H[(size_t)row * n + col] = -red[t] + blockdiag_at<FORM>(dacc, W, pi, c, r);
        H[(size_t)col * n + row] = val;
        continue;
      }
    }
    H[(size_t)col * n + row] = val;
    H[(size_t)row * n + col] = val;
  }
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
    const int i = (int)(t / 6), k = (int)(t - 6 * (t / 6));
    g[t] = dacc[(size_t)k * W + i];
  }
}

void launch_assemble(hipStream_t s, int form, const double *red, long dacc_off, const int *tileIJ, int ntiles, int W,
                     double *H, double *g, const double *r_in, double *r_out) {
  long total = (long)ntiles * TILE_ELEMS;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (form == 0)
    hipLaunchKernelGGL(k_assemble<0>, dim3(grid), dim3(256), 0, s, red, dacc_off, tileIJ, ntiles, W, H, g, r_in, r_out);
  else
    hipLaunchKernelGGL(k_assemble<1>, dim3(grid), dim3(256), 0, s, red, dacc_off, tileIJ, ntiles, W, H, g, r_in, r_out);
}

// (Measured and rejected, round 4, profiles/r04h_ticket_fusions.txt: K4 as ONE launch on contexts without a transport -- the tile workgroups
// write H themselves, the workgroup that draws the last ticket adds the block diagonal and writes g -- bit-identical and 0.294 instead of
// 0.055 ms at config 2, 0.185 instead of 0.026 on the shipped window: a ticket is a device-scope release per workgroup, i.e. a write-back
// of that XCD's L2, thousands of times per launch.  The same pattern in k_feature_eigen cost 0.02 ms at 196 workgroups.)

}  // namespace balm
