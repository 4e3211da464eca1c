// ORACLE -- TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Dependency-free CPU restatement (C++14, FP64) of BALM 2.0's second-order BA hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
// the checker / the timed CPU baseline.  The product (balm_amd/, libbalm_hip.so) never links it.
//
// PARITY PINNING: the reference ships no tests, golden vectors or fixed seeds, and
// its arithmetic primitives live in Eigen (un-vendored; README recommends 3.3.7), which is absent
// here.  This restatement is therefore pinned by (i) finite-difference checks of gradient/Hessian
// against the residual it restates, (ii) the exact rank-3 + block-diagonal identity, (iii) the
// right<->left adjoint relation, (iv) an independent numpy/LAPACK twin (oracle/numpy_oracle.py),
// and (v) where oracle/_ref builds, the reference's own source compiled against a minimal
// Eigen/PCL stand-in (oracle/compat/).  Eigen's SelfAdjointEigenSolver and LDLT are restated
// here (cyclic Jacobi; diagonal-pivot LDLT following Eigen's algorithm), not linked:
// "parity unpinned" at the Eigen boundary.
//
// Reference files restated (paths relative to /root/reference):
//   include/tools.hpp:56-71 (Exp) :92-97 (Log) :99-106 (hat) :290-349 (PointCluster)
//   src/benchmark/bavoxel.hpp:53-158 (acc_evaluate2, right form)
//   src/benchmark/bavoxel.hpp:304-426 + benchmark_virtual.cpp:218-348 (left_evaluate_acc2)
//   src/benchmark/bavoxel.hpp:428-470 (evaluate_only_residual)
//   src/benchmark/bavoxel.hpp:1025-1059 (divide_thread_left: feature split over threads)
//   src/benchmark/bavoxel.hpp:1069-1166 + benchmark_virtual.cpp:375-482 (LM damping loop)
//   src/benchmark/benchmark_virtual.cpp:48-61 (rsme)
#pragma once
#include <cmath>
#include <cstring>
#include <cfloat>
#include <vector>
#include <thread>
#include <algorithm>

namespace orc {

// ---- flat layouts shared with the product C ABI (include/balm_hip.h) -------------------------
// cluster (10 doubles): Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N          (tools.hpp:290-295)
// pose    (12 doubles): R column-major (R(r,c) = q[3*c+r]), then p  (tools.hpp:144-145)
constexpr int CL = 10;
constexpr int PS = 12;

template <int R, int C>
struct Mat {
  double a[R * C];
  double &operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  void zero() { for (int i = 0; i < R * C; i++) a[i] = 0.0; }
};
typedef Mat<3, 3> M3;
typedef Mat<4, 4> M4;
typedef Mat<3, 1> V3;
typedef Mat<6, 1> V6;
typedef Mat<6, 6> M6;

template <int R, int K, int C>
inline Mat<R, C> mul(const Mat<R, K> &x, const Mat<K, C> &y) {
  Mat<R, C> o;
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) {
      double s = 0;
      for (int k = 0; k < K; k++) s += x(r, k) * y(k, c);
      o(r, c) = s;
    }
  return o;
}
template <int R, int C>
inline Mat<C, R> tr(const Mat<R, C> &x) {
  Mat<C, R> o;
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) o(c, r) = x(r, c);
  return o;
}
template <int R, int C>
inline Mat<R, C> add(const Mat<R, C> &x, const Mat<R, C> &y) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; i++) o.a[i] = x.a[i] + y.a[i];
  return o;
}
template <int R, int C>
inline Mat<R, C> sub(const Mat<R, C> &x, const Mat<R, C> &y) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; i++) o.a[i] = x.a[i] - y.a[i];
  return o;
}
template <int R, int C>
inline Mat<R, C> scl(const Mat<R, C> &x, double s) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; i++) o.a[i] = x.a[i] * s;
  return o;
}
inline M3 eye3() {
  M3 o; o.zero(); o(0, 0) = o(1, 1) = o(2, 2) = 1.0; return o;
}
inline V3 vec3(double x, double y, double z) { V3 o; o.a[0] = x; o.a[1] = y; o.a[2] = z; return o; }
inline double dot3(const V3 &x, const V3 &y) { return x.a[0] * y.a[0] + x.a[1] * y.a[1] + x.a[2] * y.a[2]; }
inline double norm3(const V3 &x) { return std::sqrt(dot3(x, x)); }

// tools.hpp:99-106
inline M3 hat(const V3 &v) {
  M3 o; o.zero();
  o(0, 1) = -v.a[2]; o(0, 2) = v.a[1];
  o(1, 0) = v.a[2];  o(1, 2) = -v.a[0];
  o(2, 0) = -v.a[1]; o(2, 1) = v.a[0];
  return o;
}

// tools.hpp:56-71  Rodrigues, identity below |w| < 1e-11
inline M3 Exp(const V3 &w) {
  double n = norm3(w);
  if (n >= 1e-11) {
    M3 K = hat(vec3(w.a[0] / n, w.a[1] / n, w.a[2] / n));     // `ang / ang_norm`: a division
    // evaluation order of `I33 + sin*K + (1-cos)*K*K`: ((1-cos)*K)*K
    return add(add(eye3(), scl(K, std::sin(n))), mul(scl(K, 1.0 - std::cos(n)), K));
  }
  return eye3();
}

// tools.hpp:92-97
inline V3 Log(const M3 &R) {
  double trc = R(0, 0) + R(1, 1) + R(2, 2);
  double theta = (trc > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (trc - 1));
  V3 K = vec3(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
  return (std::fabs(theta) < 0.001) ? scl(K, 0.5) : scl(K, 0.5 * theta / std::sin(theta));
}

struct Pose { M3 R; V3 p; };
inline Pose load_pose(const double *q) {
  Pose x;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) x.R(r, c) = q[3 * c + r];
  x.p = vec3(q[9], q[10], q[11]);
  return x;
}
inline void store_pose(const Pose &x, double *q) {
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) q[3 * c + r] = x.R(r, c);
  q[9] = x.p.a[0]; q[10] = x.p.a[1]; q[11] = x.p.a[2];
}

// tools.hpp:290-349
struct Cluster {
  M3 P; V3 v; double N;   // the reference keeps N as int; values are integral
  Cluster() { P.zero(); v.zero(); N = 0; }
  void push(const V3 &q) {            // tools.hpp:311-316
    N += 1;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P(r, c) += q.a[r] * q.a[c];
    for (int r = 0; r < 3; r++) v.a[r] += q.a[r];
  }
  void operator+=(const Cluster &o) { // tools.hpp:324-331
    P = add(P, o.P); v = add(v, o.v); N += o.N;
  }
  // tools.hpp:333-339
  void transform(const Cluster &s, const Pose &x) {
    N = s.N;
    V3 Rv = mul(x.R, s.v);
    v = add(Rv, scl(x.p, N));
    Mat<3, 3> rp = mul(Rv, tr(x.p));
    P = add(add(add(mul(mul(x.R, s.P), tr(x.R)), rp), tr(rp)), scl(mul(x.p, tr(x.p)), N));
  }
  M3 cov() const {                    // tools.hpp:318-322
    V3 c = scl(v, 1.0 / N);
    return sub(scl(P, 1.0 / N), mul(c, tr(c)));
  }
};
inline Cluster load_cluster(const double *q) {
  Cluster s;
  s.P(0, 0) = q[0]; s.P(0, 1) = s.P(1, 0) = q[1]; s.P(0, 2) = s.P(2, 0) = q[2];
  s.P(1, 1) = q[3]; s.P(1, 2) = s.P(2, 1) = q[4]; s.P(2, 2) = q[5];
  s.v = vec3(q[6], q[7], q[8]); s.N = q[9];
  return s;
}
inline void store_cluster(const Cluster &s, double *q) {
  q[0] = s.P(0, 0); q[1] = s.P(0, 1); q[2] = s.P(0, 2); q[3] = s.P(1, 1); q[4] = s.P(1, 2);
  q[5] = s.P(2, 2); q[6] = s.v.a[0]; q[7] = s.v.a[1]; q[8] = s.v.a[2]; q[9] = s.N;
}

// Stand-in for Eigen::SelfAdjointEigenSolver<Matrix3d> (bavoxel.hpp:79,345,452): cyclic Jacobi,
// eigenvalues ascending, eigenvectors in the columns of U.  All downstream terms are even in
// each eigenvector, so the sign convention is irrelevant.
inline void eig3(const M3 &Ain, double lam[3], M3 &U) {
  M3 A = Ain;
  U = eye3();
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
    double dia = A(0, 0) * A(0, 0) + A(1, 1) * A(1, 1) + A(2, 2) * A(2, 2);
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A(p, q);
        if (apq == 0.0) continue;
        double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {   // A <- A J
          double akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq; A(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {   // A <- J^T A
          double apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk; A(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {   // U <- U J
          double ukp = U(k, p), ukq = U(k, q);
          U(k, p) = c * ukp - s * ukq; U(k, q) = s * ukp + c * ukq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  double d[3] = {A(0, 0), A(1, 1), A(2, 2)};
  std::sort(idx, idx + 3, [&](int x, int y) { return d[x] < d[y]; });
  M3 Us;
  for (int k = 0; k < 3; k++) {
    lam[k] = d[idx[k]];
    for (int r = 0; r < 3; r++) Us(r, k) = U(r, idx[k]);
  }
  U = Us;
}

struct Problem {
  int W, F;
  const double *clusters;  // F*W*CL
  const double *fix;       // F*CL or nullptr  (benchmark_virtual.cpp:241-243)
  const double *coeffs;    // F
};

// bavoxel.hpp:428-470 / benchmark_virtual.cpp:350-373
inline double only_residual(const Problem &pb, const double *poses) {
  std::vector<Pose> xs(pb.W);
  for (int i = 0; i < pb.W; i++) xs[i] = load_pose(poses + PS * i);
  double residual = 0;
  for (int a = 0; a < pb.F; a++) {
    Cluster sig;
    if (pb.fix) sig = load_cluster(pb.fix + CL * a);
    for (int i = 0; i < pb.W; i++) {
      Cluster so = load_cluster(pb.clusters + ((size_t)a * pb.W + i) * CL);
      if (so.N != 0) { Cluster st; st.transform(so, xs[i]); sig += st; }
    }
    double lam[3]; M3 U;
    eig3(sig.cov(), lam, U);
    residual += pb.coeffs[a] * lam[0];
  }
  return residual;
}

// LEFT form.  bavoxel.hpp:304-426, with C initialised from the fix cluster as
// benchmark_virtual.cpp:241-243 does.  Hess is n x n column-major (symmetric, so order-free),
// n = 6W; outputs are overwritten; features [head,end).
inline void left_evaluate(const Problem &pb, const double *poses, int head, int end,
                          double *Hess, double *JacT, double *residual) {
  const int W = pb.W, n = 6 * W;
  std::memset(Hess, 0, sizeof(double) * (size_t)n * n);
  std::memset(JacT, 0, sizeof(double) * n);
  *residual = 0;
  std::vector<Pose> xs(W);
  std::vector<M4> T(W);
  for (int i = 0; i < W; i++) {
    xs[i] = load_pose(poses + PS * i);
    T[i].zero();
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) T[i](r, c) = xs[i].R(r, c);
      T[i](r, 3) = xs[i].p.a[r];
    }
    T[i](3, 3) = 1.0;
  }
  std::vector<M4> TC(W), TCT(W);
  std::vector<int> Ns(W);
  std::vector<V6> gk[3];
  for (int k = 0; k < 3; k++) gk[k].resize(W);
  std::vector<V6> wv(W);
  auto H = [&](int r, int c) -> double & { return Hess[(size_t)c * n + r]; };

  for (int a = head; a < end; a++) {
    const double coe = pb.coeffs[a];
    M4 C; C.zero();
    if (pb.fix) {
      Cluster f = load_cluster(pb.fix + CL * a);
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) C(r, c) = f.P(r, c);
        C(r, 3) = C(3, r) = f.v.a[r];
      }
      C(3, 3) = f.N;
    }
    for (int j = 0; j < W; j++) {
      Cluster s = load_cluster(pb.clusters + ((size_t)a * W + j) * CL);
      Ns[j] = 0;
      if ((int)s.N > 0) {
        M4 Co;
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) Co(r, c) = s.P(r, c);
          Co(r, 3) = Co(3, r) = s.v.a[r];
        }
        Co(3, 3) = s.N;
        TC[j] = mul(T[j], Co);
        TCT[j] = mul(TC[j], tr(T[j]));
        C = add(C, TCT[j]);
        Ns[j] = (int)s.N;
      }
    }
    const double NN = C(3, 3);
    C = scl(C, 1.0 / NN);
    V3 vbar = vec3(C(0, 3), C(1, 3), C(2, 3));
    M3 cov;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) cov(r, c) = C(r, c) - vbar.a[r] * vbar.a[c];
    double lam[3]; M3 Uev;
    eig3(cov, lam, Uev);
    *residual += coe * lam[0];

    V3 u[3];
    Mat<6, 4> U[3];
    for (int k = 0; k < 3; k++) {
      u[k] = vec3(Uev(0, k), Uev(1, k), Uev(2, k));
      U[k].zero();
      M3 hm = hat(scl(u[k], -1.0));
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) U[k](r, c) = hm(r, c);
      for (int r = 0; r < 3; r++) U[k](3 + r, 3) = u[k].a[r];
    }

    for (int i = 0; i < W; i++) {
      if (Ns[i] == 0) continue;
      Mat<3, 4> tmp;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) tmp(r, c) = T[i](r, c);
        tmp(r, 3) = T[i](r, 3) - vbar.a[r];
      }
      Mat<4, 3> M = mul(TC[i], tr(tmp));
      for (int k = 0; k < 3; k++) {
        V6 g1 = mul(U[k], mul(M, u[0]));
        V6 g2 = mul(U[0], mul(M, u[k]));
        gk[k][i] = scl(add(g1, g2), 1.0 / NN);
      }
      Mat<6, 4> UT = mul(U[0], TC[i]);
      for (int r = 0; r < 6; r++) wv[i].a[r] = UT(r, 3);
      for (int r = 0; r < 6; r++) JacT[6 * i + r] += coe * gk[0][i].a[r];

      M6 Ha = scl(mul(wv[i], tr(wv[i])), -2.0 / NN / NN);
      M3 M33;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M33(r, c) = M(r, c);
      M3 Ell = scl(mul(hat(mul(M33, u[0])), hat(u[0])), 1.0 / NN);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Ha(r, c) += Ell(r, c) + Ell(c, r);
      for (int k = 1; k < 3; k++)
        Ha = add(Ha, scl(mul(gk[k][i], tr(gk[k][i])), 2.0 / (lam[0] - lam[k])));
      M6 Hb = mul(mul(U[0], TCT[i]), tr(U[0]));
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++)
          H(6 * i + r, 6 * i + c) += coe * Ha(r, c) + 2.0 / NN * coe * Hb(r, c);
    }
    for (int i = 0; i < W - 1; i++) {
      if (Ns[i] == 0) continue;
      for (int j = i + 1; j < W; j++) {
        if (Ns[j] == 0) continue;
        M6 Ha = scl(mul(wv[i], tr(wv[j])), -2.0 / NN / NN);
        for (int k = 1; k < 3; k++)
          Ha = add(Ha, scl(mul(gk[k][i], tr(gk[k][j])), 2.0 / (lam[0] - lam[k])));
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) H(6 * i + r, 6 * j + c) += coe * Ha(r, c);
      }
    }
  }
  for (int i = 1; i < W; i++)       // bavoxel.hpp:422-424
    for (int j = 0; j < i; j++)
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) H(6 * i + r, 6 * j + c) = H(6 * j + c, 6 * i + r);
}

// RIGHT form.  bavoxel.hpp:53-158 (derivation: src/benchmark/"Right update.pdf").
inline void right_evaluate(const Problem &pb, const double *poses, int head, int end,
                           double *Hess, double *JacT, double *residual) {
  const int W = pb.W, n = 6 * W;
  std::memset(Hess, 0, sizeof(double) * (size_t)n * n);
  std::memset(JacT, 0, sizeof(double) * n);
  *residual = 0;
  std::vector<Pose> xs(W);
  for (int i = 0; i < W; i++) xs[i] = load_pose(poses + PS * i);
  std::vector<Cluster> so(W);
  std::vector<V3> a_i(W);           // viRiTuk
  std::vector<M3> a_uk(W);          // viRiTukukT
  std::vector<Mat<3, 6>> Auk(W);
  auto H = [&](int r, int c) -> double & { return Hess[(size_t)c * n + r]; };

  for (int a = head; a < end; a++) {
    const double coe = pb.coeffs[a];
    Cluster sig;
    if (pb.fix) sig = load_cluster(pb.fix + CL * a);
    for (int i = 0; i < W; i++) {
      so[i] = load_cluster(pb.clusters + ((size_t)a * W + i) * CL);
      if (so[i].N != 0) { Cluster st; st.transform(so[i], xs[i]); sig += st; }
    }
    V3 vbar = scl(sig.v, 1.0 / sig.N);
    double lam[3]; M3 Uev;
    eig3(sub(scl(sig.P, 1.0 / sig.N), mul(vbar, tr(vbar))), lam, Uev);
    const int NN = (int)sig.N;      // int in the reference (bavoxel.hpp:82)
    V3 u[3];
    for (int k = 0; k < 3; k++) u[k] = vec3(Uev(0, k), Uev(1, k), Uev(2, k));
    const V3 &uk = u[0];
    M3 ukukT = mul(uk, tr(uk));
    M3 umumT; umumT.zero();
    for (int m = 1; m < 3; m++) umumT = add(umumT, scl(mul(u[m], tr(u[m])), 2.0 / (lam[0] - lam[m])));

    for (int i = 0; i < W; i++) {
      if (so[i].N == 0) continue;
      const M3 &Pi = so[i].P; const V3 &vi = so[i].v; const M3 &Ri = xs[i].R;
      const double ni = so[i].N;
      M3 vihat = hat(vi);
      V3 r = mul(tr(Ri), uk);
      M3 rh = hat(r);
      V3 Pir = mul(Pi, r);
      a_i[i] = mul(vihat, r);
      a_uk[i] = mul(a_i[i], tr(uk));
      V3 ti = sub(xs[i].p, vbar);
      double s = dot3(uk, ti);
      M3 combo1 = add(hat(Pir), scl(vihat, s));
      V3 combo2 = add(mul(Ri, vi), scl(ti, ni));
      M3 left = sub(mul(add(mul(Ri, Pi), mul(ti, tr(vi))), rh), mul(Ri, combo1));
      M3 right = add(mul(combo2, tr(uk)), scl(eye3(), dot3(combo2, uk)));
      for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) {
          Auk[i](rr, c) = left(rr, c) / NN;
          Auk[i](rr, 3 + c) = right(rr, c) / NN;
        }
      V6 jjt = mul(tr(Auk[i]), uk);
      for (int rr = 0; rr < 6; rr++) JacT[6 * i + rr] += coe * jjt.a[rr];

      M3 HRt = scl(a_uk[i], 2.0 / NN * (1.0 - ni / NN));
      M6 Hb = mul(mul(tr(Auk[i]), umumT), Auk[i]);
      M3 tl = sub(sub(scl(mul(sub(combo1, mul(rh, Pi)), rh), 2.0 / NN),
                      scl(mul(a_i[i], tr(a_i[i])), 2.0 / NN / NN)),
                  scl(hat(vec3(jjt.a[0], jjt.a[1], jjt.a[2])), 0.5));
      for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) {
          Hb(rr, c) += tl(rr, c);
          Hb(rr, 3 + c) += HRt(rr, c);
          Hb(3 + rr, c) += HRt(c, rr);
          Hb(3 + rr, 3 + c) += 2.0 / NN * (ni - ni * ni / NN) * ukukT(rr, c);
        }
      for (int rr = 0; rr < 6; rr++)
        for (int c = 0; c < 6; c++) H(6 * i + rr, 6 * i + c) += coe * Hb(rr, c);
    }
    for (int i = 0; i < W - 1; i++) {
      if (so[i].N == 0) continue;
      const double ni = so[i].N;
      for (int j = i + 1; j < W; j++) {
        if (so[j].N == 0) continue;
        const double nj = so[j].N;
        M6 Hb = mul(mul(tr(Auk[i]), umumT), Auk[j]);
        M3 c00 = scl(mul(a_i[i], tr(a_i[j])), -2.0 / NN / NN);
        for (int rr = 0; rr < 3; rr++)
          for (int c = 0; c < 3; c++) {
            Hb(rr, c) += c00(rr, c);
            Hb(rr, 3 + c) += -2.0 * nj / NN / NN * a_uk[i](rr, c);
            Hb(3 + rr, c) += -2.0 * ni / NN / NN * a_uk[j](c, rr);
            Hb(3 + rr, 3 + c) += -2.0 * ni * nj / NN / NN * ukukT(rr, c);
          }
        for (int rr = 0; rr < 6; rr++)
          for (int c = 0; c < 6; c++) H(6 * i + rr, 6 * j + c) += coe * Hb(rr, c);
      }
    }
    *residual += coe * lam[0];
  }
  for (int i = 1; i < W; i++)
    for (int j = 0; j < i; j++)
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) H(6 * i + r, 6 * j + c) = H(6 * j + c, 6 * i + r);
}

// bavoxel.hpp:1025-1059: split [0,F) into `threads` real-valued parts, one std::thread each with
// thread-private outputs, serial sum after join.  threads<=1 (or F<threads) -> one part.
inline double evaluate_threads(int form, const Problem &pb, const double *poses, int threads,
                               double *Hess, double *JacT) {
  const int n = 6 * pb.W;
  int T = threads < 1 ? 1 : threads;
  if (pb.F < T) T = 1;
  auto run = [&](int head, int end, double *H, double *J, double *r) {
    if (form == 0) left_evaluate(pb, poses, head, end, H, J, r);
    else right_evaluate(pb, poses, head, end, H, J, r);
  };
  if (T == 1) { double r; run(0, pb.F, Hess, JacT, &r); return r; }
  std::vector<std::vector<double>> Hs(T), Js(T);
  std::vector<double> rs(T, 0.0);
  std::vector<std::thread> th;
  double part = 1.0 * pb.F / T;
  for (int t = 0; t < T; t++) {
    Hs[t].resize((size_t)n * n); Js[t].resize(n);
    th.emplace_back(run, (int)(part * t), (int)(part * (t + 1)), Hs[t].data(), Js[t].data(), &rs[t]);
  }
  std::memset(Hess, 0, sizeof(double) * (size_t)n * n);
  std::memset(JacT, 0, sizeof(double) * n);
  double residual = 0;
  for (int t = 0; t < T; t++) {
    th[t].join();
    for (size_t k = 0; k < (size_t)n * n; k++) Hess[k] += Hs[t][k];
    for (int k = 0; k < n; k++) JacT[k] += Js[t][k];
    residual += rs[t];
  }
  return residual;
}

// Stand-in for Eigen::LDLT (bavoxel.hpp:1114 `.ldlt().solve()`), following Eigen 3.3's in-place
// lower algorithm: at step k the pivot is the largest |diagonal| among rows k..n-1 (the trailing
// diagonal is NOT yet updated when it is searched -- the update is left-looking), symmetric swap,
// then a_kk -= A10 D A10^T, A21 -= A20 D A10^T, A21 /= a_kk.  D may be negative (indefinite A).
// solve = P^T L^-T D^+ L^-1 P b, with D^+ zeroing |d| <= DBL_MIN.  A: n x n column-major (only
// the lower triangle is read); returns the number of negative pivots.
inline int ldlt_solve(int n, const double *Ain, const double *b, double *x) {
  std::vector<double> A((size_t)n * n);
  std::memcpy(A.data(), Ain, sizeof(double) * (size_t)n * n);
  auto M = [&](int r, int c) -> double & { return A[(size_t)c * n + r]; };
  std::vector<int> tp(n);
  std::vector<double> temp(n);
  int neg = 0;
  for (int k = 0; k < n; k++) {
    int big = k; double bv = std::fabs(M(k, k));
    for (int j = k + 1; j < n; j++) if (std::fabs(M(j, j)) > bv) { bv = std::fabs(M(j, j)); big = j; }
    tp[k] = big;
    if (big != k) {
      for (int c = 0; c < k; c++) std::swap(M(k, c), M(big, c));
      for (int r = big + 1; r < n; r++) std::swap(M(r, k), M(r, big));
      std::swap(M(k, k), M(big, big));
      for (int i = k + 1; i < big; i++) std::swap(M(i, k), M(big, i));
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int c = 0; c < k; c++) temp[c] = M(c, c) * M(k, c);
      double s = 0;
      for (int c = 0; c < k; c++) s += M(k, c) * temp[c];
      M(k, k) -= s;
      for (int c = 0; c < k; c++) {
        const double t = temp[c];
        if (t == 0.0) continue;
        const double *col = &A[(size_t)c * n + k + 1];
        double *dst = &A[(size_t)k * n + k + 1];
        for (int r = 0; r < rs; r++) dst[r] -= col[r] * t;
      }
    }
    const double akk = M(k, k);
    if (akk < 0) neg++;
    if (rs > 0 && std::fabs(akk) > 0.0) {
      double *dst = &A[(size_t)k * n + k + 1];
      for (int r = 0; r < rs; r++) dst[r] /= akk;
    }
  }
  std::vector<double> y(b, b + n);
  for (int k = 0; k < n; k++) std::swap(y[k], y[tp[k]]);           // P b
  for (int c = 0; c < n; c++) {                                     // L^-1
    const double yc = y[c];
    for (int r = c + 1; r < n; r++) y[r] -= M(r, c) * yc;
  }
  for (int k = 0; k < n; k++) {                                     // D^+
    const double d = M(k, k);
    y[k] = (std::fabs(d) > DBL_MIN) ? y[k] / d : 0.0;
  }
  for (int c = n - 1; c >= 0; c--) {                                // L^-T
    double s = y[c];
    for (int r = c + 1; r < n; r++) s -= M(r, c) * y[r];
    y[c] = s;
  }
  for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[tp[k]]);       // P^T
  std::memcpy(x, y.data(), sizeof(double) * n);
  return neg;
}

struct IterLog { double r1, r2, u, v, q, q1; int accepted; int hess_evaluated; };

// (H + u diag H) dx = -g ; q1 = 0.5 dx.(u D dx - g)     bavoxel.hpp:1113-1114,1127
inline void solve_damped(int n, const double *Hess, const double *JacT, double u, double *dxi, double *q1) {
  std::vector<double> A((size_t)n * n), nb(n);
  std::memcpy(A.data(), Hess, sizeof(double) * (size_t)n * n);
  for (int k = 0; k < n; k++) { A[(size_t)k * n + k] += u * Hess[(size_t)k * n + k]; nb[k] = -JacT[k]; }
  ldlt_solve(n, A.data(), nb.data(), dxi);
  double s = 0;
  for (int k = 0; k < n; k++) s += dxi[k] * (u * Hess[(size_t)k * n + k] * dxi[k] - JacT[k]);
  *q1 = 0.5 * s;
}

// pose update: form 0 (left) bavoxel.hpp:1123-1125 ; form 1 (right) bavoxel.hpp:1119-1120
inline void update_poses(int form, int W, const double *poses, const double *dxi, double *out) {
  for (int j = 0; j < W; j++) {
    Pose x = load_pose(poses + PS * j), y;
    M3 dR = Exp(vec3(dxi[6 * j], dxi[6 * j + 1], dxi[6 * j + 2]));
    V3 dt = vec3(dxi[6 * j + 3], dxi[6 * j + 4], dxi[6 * j + 5]);
    if (form == 0) { y.R = mul(dR, x.R); y.p = add(mul(dR, x.p), dt); }
    else           { y.R = mul(x.R, dR); y.p = add(x.p, dt); }
    store_pose(y, out + PS * j);
  }
}

// bavoxel.hpp:1159-1164
inline void reanchor(int W, double *poses) {
  Pose e0 = load_pose(poses);
  for (int j = 0; j < W; j++) {
    Pose x = load_pose(poses + PS * j), y;
    y.p = mul(tr(e0.R), sub(x.p, e0.p));
    y.R = mul(tr(e0.R), x.R);
    store_pose(y, poses + PS * j);
  }
}

// LM loop: bavoxel.hpp:1087-1164 (u0 = 0.01, max_iter = 10) / benchmark_virtual.cpp:380-479
// (u0 = 0.1, max_iter = 20).  Returns iterations executed; poses updated in place, re-anchored.
inline int damping_iter(int form, const Problem &pb, double *poses, double u0, int max_iter,
                        double rel_tol, int threads, IterLog *log) {
  const int W = pb.W, n = 6 * W;
  double u = u0, v = 2;
  std::vector<double> Hess((size_t)n * n), JacT(n), dxi(n), xt(PS * W);
  double r1 = 0, r2 = 0, q;
  bool calc = true;
  int it = 0;
  for (; it < max_iter;) {
    const bool evaluated = calc;
    if (calc) r1 = evaluate_threads(form, pb, poses, threads, Hess.data(), JacT.data());
    double q1;
    solve_damped(n, Hess.data(), JacT.data(), u, dxi.data(), &q1);
    update_poses(form, W, poses, dxi.data(), xt.data());
    r2 = only_residual(pb, xt.data());
    q = r1 - r2;
    if (log) { log[it].r1 = r1; log[it].r2 = r2; log[it].u = u; log[it].v = v; log[it].q1 = q1;
               log[it].q = q; log[it].accepted = q > 0; log[it].hess_evaluated = evaluated; }
    if (q > 0) {
      std::memcpy(poses, xt.data(), sizeof(double) * PS * W);
      q = q / q1; v = 2; q = 1 - std::pow(2 * q - 1, 3);
      u *= (q < 1.0 / 3.0 ? 1.0 / 3.0 : q);
      calc = true;
    } else {
      u = u * v; v = 2 * v; calc = false;
    }
    it++;
    if (std::fabs(r1 - r2) / r1 < rel_tol) break;
  }
  reanchor(W, poses);
  return it;
}

// benchmark_virtual.cpp:48-61
inline void rsme(int W, const double *gt, const double *es, double *rot, double *tran) {
  double r = 0, t = 0;
  for (int i = 0; i < W; i++) {
    Pose g = load_pose(gt + PS * i), e = load_pose(es + PS * i);
    V3 l = Log(mul(tr(g.R), e.R));
    r += dot3(l, l);
    V3 d = sub(e.p, g.p);
    t += dot3(d, d);
  }
  *rot = std::sqrt(r / W); *tran = std::sqrt(t / W);
}

}  // namespace orc
