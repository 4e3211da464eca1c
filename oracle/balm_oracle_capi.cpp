// ORACLE -- TEST INFRASTRUCTURE ONLY (see balm_oracle.hpp).  Plain-C entry points so that
// tests/, smoke() and bench.py's cpu_baseline leg can drive the CPU restatement through ctypes.
#include "balm_oracle.hpp"
#include <chrono>

using namespace orc;

extern "C" {

void orc_exp(const double *w, double *R9) {
  M3 R = Exp(vec3(w[0], w[1], w[2]));
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R9[3 * c + r] = R(r, c);
}

void orc_log(const double *R9, double *w) {
  M3 R;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R(r, c) = R9[3 * c + r];
  V3 l = Log(R);
  w[0] = l.a[0]; w[1] = l.a[1]; w[2] = l.a[2];
}

// eigenvalues ascending, U column-major (column k = eigenvector k)
void orc_eig3(const double *A9, double *lam, double *U9) {
  M3 A, U;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) A(r, c) = A9[3 * c + r];
  eig3(A, lam, U);
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) U9[3 * c + r] = U(r, c);
}

// xyz: n*3 doubles -> cluster (10 doubles)             tools.hpp:311-316
void orc_cluster_push(const double *xyz, long n, double *cl) {
  Cluster s = load_cluster(cl);
  for (long k = 0; k < n; k++) s.push(vec3(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]));
  store_cluster(s, cl);
}

void orc_cluster_transform(const double *cl, const double *pose, double *out) {
  Cluster s = load_cluster(cl), t;
  t.transform(s, load_pose(pose));
  store_cluster(t, out);
}

int orc_evaluate(int form, int W, int F, const double *clusters, const double *fix,
                 const double *coeffs, const double *poses, int head, int end, double *Hess,
                 double *JacT, double *residual) {
  Problem pb{W, F, clusters, fix, coeffs};
  if (head < 0 || end > F || head > end) return 1;
  if (form == 0) left_evaluate(pb, poses, head, end, Hess, JacT, residual);
  else if (form == 1) right_evaluate(pb, poses, head, end, Hess, JacT, residual);
  else return 2;
  return 0;
}

double orc_evaluate_threads(int form, int W, int F, const double *clusters, const double *fix,
                            const double *coeffs, const double *poses, int threads, double *Hess,
                            double *JacT) {
  Problem pb{W, F, clusters, fix, coeffs};
  return evaluate_threads(form, pb, poses, threads, Hess, JacT);
}

double orc_only_residual(int W, int F, const double *clusters, const double *fix,
                         const double *coeffs, const double *poses) {
  Problem pb{W, F, clusters, fix, coeffs};
  return only_residual(pb, poses);
}

int orc_ldlt_solve(int n, const double *A, const double *b, double *x) {
  return ldlt_solve(n, A, b, x);
}

void orc_solve_damped(int n, const double *Hess, const double *JacT, double u, double *dxi,
                      double *q1) {
  solve_damped(n, Hess, JacT, u, dxi, q1);
}

void orc_update_poses(int form, int W, const double *poses, const double *dxi, double *out) {
  update_poses(form, W, poses, dxi, out);
}

void orc_reanchor(int W, double *poses) { reanchor(W, poses); }

// log: max_iter rows of 8 doubles: r1 r2 u v q q1 accepted hess_evaluated
int orc_damping_iter(int form, int W, int F, const double *clusters, const double *fix,
                     const double *coeffs, double *poses, double u0, int max_iter, double rel_tol,
                     int threads, double *log8) {
  Problem pb{W, F, clusters, fix, coeffs};
  std::vector<IterLog> lg(max_iter);
  int it = damping_iter(form, pb, poses, u0, max_iter, rel_tol, threads, lg.data());
  if (log8)
    for (int k = 0; k < it; k++) {
      double *o = log8 + 8 * k;
      o[0] = lg[k].r1; o[1] = lg[k].r2; o[2] = lg[k].u; o[3] = lg[k].v; o[4] = lg[k].q;
      o[5] = lg[k].q1; o[6] = lg[k].accepted; o[7] = lg[k].hess_evaluated;
    }
  return it;
}

void orc_rsme(int W, const double *gt, const double *es, double *rot, double *tran) {
  rsme(W, gt, es, rot, tran);
}

// CPU-baseline timing leg: one LM-iteration's worth of work on features [0, F_sample) of the
// workload -- one Hessian evaluation (threads as given), and one residual-only evaluation.
// out[0] = seconds(evaluate), out[1] = seconds(only_residual).  The dense solve is timed
// separately by orc_time_solve (it does not scale with F).
void orc_time_sample(int form, int W, int F_sample, const double *clusters, const double *fix,
                     const double *coeffs, const double *poses, int threads, double *out) {
  Problem pb{W, F_sample, clusters, fix, coeffs};
  const int n = 6 * W;
  std::vector<double> H((size_t)n * n), J(n);
  auto t0 = std::chrono::steady_clock::now();
  volatile double r = evaluate_threads(form, pb, poses, threads, H.data(), J.data());
  auto t1 = std::chrono::steady_clock::now();
  volatile double r2 = only_residual(pb, poses);
  auto t2 = std::chrono::steady_clock::now();
  (void)r; (void)r2;
  out[0] = std::chrono::duration<double>(t1 - t0).count();
  out[1] = std::chrono::duration<double>(t2 - t1).count();
}

double orc_time_solve(int n, const double *Hess, const double *JacT, double u) {
  std::vector<double> dx(n); double q1;
  auto t0 = std::chrono::steady_clock::now();
  solve_damped(n, Hess, JacT, u, dx.data(), &q1);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
