// ORACLE -- TEST INFRASTRUCTURE ONLY.
// Compiles the REFERENCE'S OWN consistency / covariance sources where they lie (never copied):
//     /root/reference/src/simulation/toolss.hpp    (PointCluster with its 9x9 noise covariance c_cov)
//     /root/reference/src/simulation/BAs_left.hpp  (VOX_HESS::left_jacobian_point, BALM2::multi_second,
//                                                   the `Rcov = H^-1 Rcov H^-T` tail of damping_iter)
// against the stand-in headers in oracle/compat/ and exposes them with the flat layouts of include/balm_hip.h.
// A separate shared object (oracle/_ref/libbalm_ref_sim.so) because these headers re-declare the class names
// of src/benchmark/bavoxel.hpp.  Pins oracle/balm_oracle.hpp's covariance restatement ("next" row N4).
#include <ros/ros.h>

#include <cstdio>
#include <cstring>

#include "toolss.hpp"
#include "BAs_left.hpp"

namespace {

struct Problem {
  std::vector<std::vector<PointCluster> *> feats;
  std::vector<PointCluster *> fixes;
  VOX_HESS vh;
  ~Problem() {
    for (auto p : feats) delete p;
    for (auto p : fixes) delete p;
  }
};

PointCluster make_cluster(const double *q, const double *ccov) {
  PointCluster c;
  c.P << q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5];
  c.v << q[6], q[7], q[8];
  c.N = (int)q[9];
  if (ccov)
    for (int r = 0; r < 9; r++) for (int k = 0; k < 9; k++) c.c_cov(r, k) = ccov[9 * r + k];
  return c;
}

void build(Problem &pb, int W, int F, const double *clusters, const double *ccov, const double *fix) {
  win_size = W;       // BAs_left.hpp:13
  for (int a = 0; a < F; a++) {
    auto *v = new std::vector<PointCluster>(W);
    for (int i = 0; i < W; i++)
      (*v)[i] = make_cluster(clusters + ((size_t)a * W + i) * 10, ccov ? ccov + ((size_t)a * W + i) * 81 : nullptr);
    PointCluster *fx = new PointCluster();
    if (fix) *fx = make_cluster(fix + (size_t)a * 10, nullptr);
    pb.feats.push_back(v);
    pb.fixes.push_back(fx);
    pb.vh.plvec_voxels.push_back(v);
    pb.vh.sig_vecs.push_back(fx);
    pb.vh.coeffs.push_back(1.0);        // push_voxel: `coe = 1` (BAs_left.hpp:44)
  }
}

std::vector<IMUST> load_poses(int W, const double *poses) {
  std::vector<IMUST> xs(W);
  for (int i = 0; i < W; i++) {
    const double *q = poses + 12 * i;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) xs[i].R(r, c) = q[3 * c + r];
    xs[i].p << q[9], q[10], q[11];
  }
  return xs;
}

struct Quiet {          // left_jacobian_point prints a progress line
  FILE *saved;
  Quiet() { fflush(stdout); saved = stdout; stdout = fopen("/dev/null", "w"); }
  ~Quiet() { fclose(stdout); stdout = saved; }
};

}  // namespace

extern "C" {

// PointCluster::push with POINT_NOISE (toolss.hpp:315-347): cluster[10] and c_cov[81] of n points
void refsim_cluster_push(const double *pts, int n, double pn, double *cluster, double *ccov) {
  pnoise = pn;
  PointCluster c;
  for (int k = 0; k < n; k++) c.push(Eigen::Vector3d(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]));
  cluster[0] = c.P(0, 0); cluster[1] = c.P(0, 1); cluster[2] = c.P(0, 2); cluster[3] = c.P(1, 1); cluster[4] = c.P(1, 2);
  cluster[5] = c.P(2, 2); cluster[6] = c.v[0]; cluster[7] = c.v[1]; cluster[8] = c.v[2]; cluster[9] = c.N;
  for (int r = 0; r < 9; r++) for (int k = 0; k < 9; k++) ccov[9 * r + k] = c.c_cov(r, k);
}

// VOX_HESS::left_jacobian_point (BAs_left.hpp:342-473): Rcov_raw = sum Ls c_cov Ls^T over features [beg, end)
int refsim_point_cov(int W, int F, const double *clusters, const double *ccov, const double *fix, const double *poses,
                     int beg, int end, double *Rraw) {
  Problem pb;
  build(pb, W, F, clusters, ccov, fix);
  std::vector<IMUST> xs = load_poses(W, poses);
  Eigen::MatrixXd R(6 * W, 6 * W);
  {
    Quiet q;
    pb.vh.left_jacobian_point(xs, beg, end, R);
  }
  std::memcpy(Rraw, R.data(), sizeof(double) * 36 * W * W);
  return 0;
}

// the covariance tail of BALM2::damping_iter (BAs_left.hpp:1089-1096): divide_thread, multi_second, H^-1 Rcov H^-T
int refsim_pose_cov(int W, int F, const double *clusters, const double *ccov, const double *fix, const double *poses,
                    double *Hess, double *Rcov) {
  Problem pb;
  build(pb, W, F, clusters, ccov, fix);
  std::vector<IMUST> xs = load_poses(W, poses), x_ab(W);
  Eigen::MatrixXd H(6 * W, 6 * W), R(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  R.setZero();
  BALM2 opt;
  {
    Quiet q;
    opt.divide_thread(xs, pb.vh, x_ab, H, J);
    opt.multi_second(xs, R, pb.vh);
  }
  Eigen::MatrixXd hess_inv = H.inverse();
  R = hess_inv * R * hess_inv.transpose();
  if (Hess) std::memcpy(Hess, H.data(), sizeof(double) * 36 * W * W);
  std::memcpy(Rcov, R.data(), sizeof(double) * 36 * W * W);
  return 0;
}

// The association of consistency.cpp:96-150 on caller-supplied scans: cut_voxel per scan (BAs_left.hpp:1102), recut
// (:713), marginalize the first `fix` scans (:754, :911), tras_opt (:794).  xyz: all scans concatenated, counts[n_scans].
// Two calls: clusters == NULL returns the number of features; then clusters [F][win][10] and fixes [F][10].
static std::vector<double> g_assoc_cl, g_assoc_fix;
int refsim_associate(int n_scans, int fix, const float *xyz, const long *counts, const double *poses, double voxel,
                     double *clusters, double *fixes) {
  if (clusters) {
    std::memcpy(clusters, g_assoc_cl.data(), g_assoc_cl.size() * sizeof(double));
    std::memcpy(fixes, g_assoc_fix.data(), g_assoc_fix.size() * sizeof(double));
    return (int)(g_assoc_fix.size() / 10);
  }
  win_size = n_scans - fix;
  fix_size = fix;
  voxel_size = voxel;
  std::vector<IMUST> x_buf = load_poses(n_scans, poses);
  std::unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> surf_map;
  long off = 0;
  for (int m = 0; m < n_scans; m++) {
    pcl::PointCloud<PointType> pl;
    for (long k = 0; k < counts[m]; k++) {
      PointType ap; ap.x = xyz[3 * (off + k)]; ap.y = xyz[3 * (off + k) + 1]; ap.z = xyz[3 * (off + k) + 2];
      pl.push_back(ap);
    }
    off += counts[m];
    cut_voxel(surf_map, pl, x_buf[m], m);
  }
  std::vector<IMUST> x_buf2;
  for (auto iter = surf_map.begin(); iter != surf_map.end(); ++iter) {
    iter->second->recut(n_scans);
    iter->second->marginalize(fix_size, x_buf2, n_scans);
  }
  VOX_HESS voxhess;
  for (auto iter = surf_map.begin(); iter != surf_map.end(); iter++) iter->second->tras_opt(voxhess, win_size);
  const size_t F = voxhess.plvec_voxels.size();
  g_assoc_cl.assign(F * win_size * 10, 0.0);
  g_assoc_fix.assign(F * 10, 0.0);
  auto put = [](const PointCluster &c, double *q) {
    q[0] = c.P(0, 0); q[1] = c.P(0, 1); q[2] = c.P(0, 2); q[3] = c.P(1, 1); q[4] = c.P(1, 2); q[5] = c.P(2, 2);
    q[6] = c.v[0]; q[7] = c.v[1]; q[8] = c.v[2]; q[9] = c.N;
  };
  for (size_t a = 0; a < F; a++) {
    for (int i = 0; i < win_size; i++) put((*voxhess.plvec_voxels[a])[i], g_assoc_cl.data() + (a * win_size + i) * 10);
    put(*voxhess.sig_vecs[a], g_assoc_fix.data() + a * 10);
  }
  for (auto &kv : surf_map) delete kv.second;
  return (int)F;
}

// VOX_HESS::left_evaluate_acc2 of the simulation copy (BAs_left.hpp), to confirm it is the same evaluator
int refsim_evaluate(int W, int F, const double *clusters, const double *fix, const double *poses, double *Hess, double *JacT,
                    double *residual) {
  Problem pb;
  build(pb, W, F, clusters, nullptr, fix);
  std::vector<IMUST> xs = load_poses(W, poses);
  Eigen::MatrixXd H(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  double r = 0;
  pb.vh.left_evaluate_acc2(xs, 0, F, H, J, r);
  std::memcpy(Hess, H.data(), sizeof(double) * 36 * W * W);
  std::memcpy(JacT, J.data(), sizeof(double) * 6 * W);
  *residual = r;
  return 0;
}

// ---- the consistency driver's map used call for call (consistency.cpp:108-136 and beyond): comparator of balm_window_* with
// the strict plane test, fix_frames and defer_recut (tests/test_gpu_window.py) ----
struct SimWin {
  std::unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> map;
  int win_count = 0;
  VOX_HESS *vh = nullptr;
  ~SimWin() { for (auto &kv : map) delete kv.second; delete vh; }
};

void *refsim_win_open(int W, int fix, double vsize) {
  win_size = W; fix_size = fix; voxel_size = vsize;
  return new SimWin();
}

void refsim_win_cut_voxel(void *hh, const float *xyz, long n, const double *pose12) {
  SimWin *h = (SimWin *)hh;
  pcl::PointCloud<PointType> pl;
  pl.reserve((size_t)n);
  for (long k = 0; k < n; k++) { PointType ap; ap.x = xyz[3 * k]; ap.y = xyz[3 * k + 1]; ap.z = xyz[3 * k + 2]; pl.push_back(ap); }
  IMUST x = load_poses(1, pose12)[0];
  cut_voxel(h->map, pl, x, h->win_count);
  h->win_count++;
}

void refsim_win_recut(void *hh) {
  SimWin *h = (SimWin *)hh;
  for (auto &kv : h->map) kv.second->recut(h->win_count);
}

void refsim_win_marginalize(void *hh, int mg, const double *poses) {
  SimWin *h = (SimWin *)hh;
  std::vector<IMUST> xs;
  if (poses) xs = load_poses(h->win_count, poses);
  for (auto &kv : h->map) kv.second->marginalize(mg, xs, h->win_count);
  h->win_count -= mg;
}

int refsim_win_features(void *hh) {
  SimWin *h = (SimWin *)hh;
  delete h->vh;
  h->vh = new VOX_HESS();
  for (auto &kv : h->map) kv.second->tras_opt(*h->vh, h->win_count);
  return (int)h->vh->plvec_voxels.size();
}

void refsim_win_export(void *hh, double *clusters, double *fix) {
  SimWin *h = (SimWin *)hh;
  const int W = win_size;
  const size_t F = h->vh->plvec_voxels.size();
  auto put = [](const PointCluster &c, double *q) {
    q[0] = c.P(0, 0); q[1] = c.P(1, 0); q[2] = c.P(2, 0); q[3] = c.P(1, 1); q[4] = c.P(2, 1); q[5] = c.P(2, 2);
    q[6] = c.v[0]; q[7] = c.v[1]; q[8] = c.v[2]; q[9] = c.N;
  };
  for (size_t a = 0; a < F; a++) {
    for (int i = 0; i < W; i++) put((*h->vh->plvec_voxels[a])[i], clusters + (a * W + i) * 10);
    put(*h->vh->sig_vecs[a], fix + a * 10);
  }
}

void refsim_win_close(void *hh) {
  delete (SimWin *)hh;
  win_size = 100; fix_size = 1; voxel_size = 1;         // the globals' defaults (BAs_left.hpp:13-23)
}

}  // extern "C"
