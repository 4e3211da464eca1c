// ORACLE -- TEST INFRASTRUCTURE ONLY.
// Compiles the REFERENCE'S OWN SOURCE FILES where they lie (never copied into this repo):
//     /root/reference/include/tools.hpp
//     /root/reference/src/benchmark/bavoxel.hpp   (VOX_HESS, BALM2, OCTO_TREE_*, cut_voxel)
// against the minimal Eigen/PCL/ROS stand-ins in oracle/compat/, and exposes them through plain-C
// entry points with the same flat layouts as include/balm_hip.h.  Built by oracle/ref_build.sh into
// oracle/_ref/libbalm_ref.so (git-ignored; travels to the GPU box).  Used to (a) pin the
// restatement in balm_oracle.hpp against the reference's literal arithmetic, (b) generate the
// golden fixtures under tests/golden/, (c) serve as bench.py's cpu_baseline kind "reference".
#include <ros/ros.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "tools.hpp"
#include "bavoxel.hpp"

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

PointCluster make_cluster(const double *q) {
  PointCluster c;
  c.P << q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5];
  c.v << q[6], q[7], q[8];
  c.N = (int)q[9];
  return c;
}

// fills VOX_HESS directly (all features kept, caller's weights), like the benchmark drivers do
void build(Problem &pb, int W, int F, const double *clusters, const double *fix, const double *coeffs) {
  win_size = W;       // the reference's global (bavoxel.hpp:17)
  for (int a = 0; a < F; a++) {
    auto *v = new std::vector<PointCluster>(W);
    for (int i = 0; i < W; i++) (*v)[i] = make_cluster(clusters + ((size_t)a * W + i) * 10);
    PointCluster *fx = new PointCluster();
    if (fix) *fx = make_cluster(fix + (size_t)a * 10);
    pb.feats.push_back(v);
    pb.fixes.push_back(fx);
    pb.vh.plvec_voxels.push_back(v);
    pb.vh.sig_vecs.push_back(fx);
    pb.vh.coeffs.push_back(coeffs[a]);
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

void store_poses(const std::vector<IMUST> &xs, double *poses) {
  for (size_t i = 0; i < xs.size(); i++) {
    double *q = poses + 12 * i;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) q[3 * c + r] = xs[i].R(r, c);
    q[9] = xs[i].p[0]; q[10] = xs[i].p[1]; q[11] = xs[i].p[2];
  }
}

}  // namespace

extern "C" {

// form 0: VOX_HESS::left_evaluate_acc2 (bavoxel.hpp:304)   form 1: VOX_HESS::acc_evaluate2 (:53)
// form 2: VOX_HESS::left_evaluate (:160, the un-accelerated left form; prints a timing line)
int ref_evaluate(int form, int W, int F, const double *clusters, const double *fix, const double *coeffs,
                 const double *poses, int head, int end, double *Hess, double *JacT, double *residual) {
  Problem pb;
  build(pb, W, F, clusters, fix, coeffs);
  std::vector<IMUST> xs = load_poses(W, poses);
  Eigen::MatrixXd H(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  double r = 0;
  if (form == 0) pb.vh.left_evaluate_acc2(xs, head, end, H, J, r);
  else if (form == 1) pb.vh.acc_evaluate2(xs, head, end, H, J, r);
  else if (form == 2) pb.vh.left_evaluate(xs, head, end, H, J, r);
  else return 1;
  std::memcpy(Hess, H.data(), sizeof(double) * 36 * W * W);
  std::memcpy(JacT, J.data(), sizeof(double) * 6 * W);
  *residual = r;
  return 0;
}

double ref_only_residual(int W, int F, const double *clusters, const double *fix, const double *coeffs,
                         const double *poses) {
  Problem pb;
  build(pb, W, F, clusters, fix, coeffs);
  std::vector<IMUST> xs = load_poses(W, poses);
  double r = 0;
  pb.vh.evaluate_only_residual(xs, r);
  return r;
}

// BALM2::divide_thread_left / divide_thread_right (bavoxel.hpp:1025, :989): 4 std::threads
double ref_divide_thread(int form, int W, int F, const double *clusters, const double *fix, const double *coeffs,
                         const double *poses, double *Hess, double *JacT) {
  Problem pb;
  build(pb, W, F, clusters, fix, coeffs);
  std::vector<IMUST> xs = load_poses(W, poses), x_ab(W);
  Eigen::MatrixXd H(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  BALM2 opt;
  double r = form == 0 ? opt.divide_thread_left(xs, pb.vh, x_ab, H, J) : opt.divide_thread_right(xs, pb.vh, x_ab, H, J);
  if (Hess) std::memcpy(Hess, H.data(), sizeof(double) * 36 * W * W);
  if (JacT) std::memcpy(JacT, J.data(), sizeof(double) * 6 * W);
  return r;
}

// `D.diagonal() = Hess.diagonal(); dxi = (Hess + u*D).ldlt().solve(-JacT);` (bavoxel.hpp:1113-1114)
void ref_solve_damped(int n, const double *Hess, const double *JacT, double u, double *dxi, double *q1) {
  Eigen::MatrixXd H(n, n), D(n, n);
  Eigen::VectorXd J(n), dx(n);
  std::memcpy(H.data(), Hess, sizeof(double) * n * n);
  std::memcpy(J.data(), JacT, sizeof(double) * n);
  D.setIdentity();
  D.diagonal() = H.diagonal();
  dx = (H + u * D).ldlt().solve(-J);
  std::memcpy(dxi, dx.data(), sizeof(double) * n);
  if (q1) *q1 = 0.5 * dx.dot(u * D * dx - J);
}

// BALM2::damping_iter (bavoxel.hpp:1069-1166).  The reference reports progress only by printf
// (:1132); stdout is captured to recover the per-iteration log (8 doubles per row:
// r1 r2 u v q q1 accepted 0).  NOTE: exits the process when a pose sees < 20 planes (:1079-1085).
int ref_damping_iter(int W, int F, const double *clusters, const double *fix, const double *coeffs, double *poses,
                     double *log8, int max_rows) {
  Problem pb;
  build(pb, W, F, clusters, fix, coeffs);
  std::vector<IMUST> xs = load_poses(W, poses);
  fflush(stdout);
  char path[] = "/tmp/balm_ref_XXXXXX";
  int fd = mkstemp(path);
  int saved = dup(1);
  dup2(fd, 1);
  BALM2 opt;
  opt.damping_iter(xs, pb.vh);
  fflush(stdout);
  dup2(saved, 1);
  close(saved);
  store_poses(xs, poses);
  int rows = 0;
  FILE *f = fdopen(fd, "r");
  rewind(f);
  char line[512];
  while (fgets(line, sizeof line, f) && rows < max_rows) {
    int it; double r1, r2, u, v, qq, q1, q;
    if (sscanf(line, "iter%d: (%lf %lf) u: %lf v: %lf q: %lf %lf %lf", &it, &r1, &r2, &u, &v, &qq, &q1, &q) == 8) {
      double *o = log8 + 8 * rows++;
      o[0] = r1; o[1] = r2; o[2] = u; o[3] = v; o[4] = q; o[5] = q1; o[6] = q > 0; o[7] = 0;
    }
  }
  fclose(f);
  unlink(path);
  return rows;
}

// VOX_HESS::push_voxel semantics (bavoxel.hpp:30-51): returns whether the feature was kept and the
// weight it was given.
int ref_push_voxel(int W, const double *clusters_a, double *coe_out) {
  win_size = W;
  std::vector<PointCluster> v(W);
  for (int i = 0; i < W; i++) v[i] = make_cluster(clusters_a + (size_t)i * 10);
  PointCluster fx;
  VOX_HESS vh;
  vh.push_voxel(&v, &fx, 0.01, 0);
  if (vh.coeffs.empty()) return 0;
  *coe_out = vh.coeffs[0];
  return 1;
}

// cpu_baseline leg: one divide_thread_left (4 threads) + one evaluate_only_residual on the first
// F_sample features.  out[0], out[1] = seconds.
void ref_time_sample(int W, int F_sample, const double *clusters, const double *coeffs, const double *poses,
                     double *out) {
  Problem pb;
  build(pb, W, F_sample, clusters, nullptr, coeffs);
  std::vector<IMUST> xs = load_poses(W, poses), x_ab(W);
  Eigen::MatrixXd H(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  BALM2 opt;
  auto t0 = std::chrono::steady_clock::now();
  volatile double r = opt.divide_thread_left(xs, pb.vh, x_ab, H, J);
  auto t1 = std::chrono::steady_clock::now();
  double r2 = 0;
  pb.vh.evaluate_only_residual(xs, r2);
  auto t2 = std::chrono::steady_clock::now();
  (void)r;
  out[0] = std::chrono::duration<double>(t1 - t0).count();
  out[1] = std::chrono::duration<double>(t2 - t1).count();
}

double ref_time_solve(int n, const double *Hess, const double *JacT, double u) {
  std::vector<double> dx(n);
  auto t0 = std::chrono::steady_clock::now();
  ref_solve_damped(n, Hess, JacT, u, dx.data(), nullptr);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---- real-world pipeline of benchmark_realworld.cpp:144-218, minus ROS/RViz ------------------------
// read_pose (:31-73) and read_file (:75-106) restated for the shipped files (alidarPose.csv: 4 text
// lines per pose, rows of [R|t], element (3,3) = timestamp; full<m>.pcd: 11 ASCII header lines ending
// "DATA binary", then 32-byte records x y z intensity normal_x normal_y normal_z curvature, all f32);
// the association itself (cut_voxel -> recut -> tras_opt -> VOX_HESS::push_voxel) is the reference's.
struct RwHandle {
  std::vector<IMUST> x_buf;
  std::unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> surf_map;
  VOX_HESS voxhess;
  long n_points = 0;
  ~RwHandle() { for (auto &kv : surf_map) delete kv.second; }
};

void *ref_rw_open(const char *dir, double vsize, int max_poses) {
  RwHandle *h = new RwHandle();
  std::string pre(dir);
  if (pre.back() != '/') pre += '/';
  FILE *f = fopen((pre + "alidarPose.csv").c_str(), "r");
  if (!f) { delete h; return nullptr; }
  std::vector<double> nums;
  double v; int ch;
  while (fscanf(f, "%lf", &v) == 1) { nums.push_back(v); do { ch = fgetc(f); } while (ch == ',' || ch == ' ' || ch == '\r' || ch == '\n'); if (ch != EOF) ungetc(ch, f); }
  fclose(f);
  int W = (int)(nums.size() / 16);
  if (max_poses > 0 && W > max_poses) W = max_poses;
  std::vector<pcl::PointCloud<PointType>::Ptr> pl_fulls;
  for (int m = 0; m < W; m++) {
    IMUST curr;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) curr.R(r, c) = nums[16 * m + 4 * r + c]; curr.p[r] = nums[16 * m + 4 * r + 3]; }
    curr.t = nums[16 * m + 15];
    h->x_buf.push_back(curr);
    pcl::PointCloud<PointType>::Ptr pl(new pcl::PointCloud<PointType>());
    FILE *pf = fopen((pre + "full" + std::to_string(m) + ".pcd").c_str(), "rb");
    if (!pf) { delete h; return nullptr; }
    char line[256]; long npts = 0;
    while (fgets(line, sizeof line, pf)) {
      if (!strncmp(line, "POINTS", 6)) npts = atol(line + 7);
      if (!strncmp(line, "DATA", 4)) break;
    }
    std::vector<float> rec((size_t)npts * 8);
    size_t got = fread(rec.data(), 32, (size_t)npts, pf);
    fclose(pf);
    pl->reserve(got);
    for (size_t k = 0; k < got; k++) { PointType ap; ap.x = rec[8 * k]; ap.y = rec[8 * k + 1]; ap.z = rec[8 * k + 2]; ap.intensity = rec[8 * k + 3]; pl->push_back(ap); }
    h->n_points += (long)got;
    pl_fulls.push_back(pl);
  }
  // benchmark_realworld.cpp:163-170
  IMUST es0 = h->x_buf[0];
  for (uint i = 0; i < h->x_buf.size(); i++) {
    h->x_buf[i].p = es0.R.transpose() * (h->x_buf[i].p - es0.p);
    h->x_buf[i].R = es0.R.transpose() * h->x_buf[i].R;
  }
  win_size = h->x_buf.size();
  voxel_size = vsize;
  // :183-200
  eigen_value_array[0] = 1.0 / 16; eigen_value_array[1] = 1.0 / 16; eigen_value_array[2] = 1.0 / 9;
  for (int i = 0; i < win_size; i++) cut_voxel(h->surf_map, *pl_fulls[i], h->x_buf[i], i);
  for (auto iter = h->surf_map.begin(); iter != h->surf_map.end(); iter++) {
    iter->second->recut(win_size);
    iter->second->tras_opt(h->voxhess, win_size);
  }
  return h;
}

void ref_rw_dims(void *hh, int *W, int *F, long *n_points) {
  RwHandle *h = (RwHandle *)hh;
  *W = (int)h->x_buf.size(); *F = (int)h->voxhess.plvec_voxels.size(); *n_points = h->n_points;
}

void ref_rw_export(void *hh, double *clusters, double *fix, double *coeffs, double *poses) {
  RwHandle *h = (RwHandle *)hh;
  const int W = (int)h->x_buf.size();
  const size_t F = h->voxhess.plvec_voxels.size();
  auto put = [](const PointCluster &c, double *q) {
    q[0] = c.P(0, 0); q[1] = c.P(0, 1); q[2] = c.P(0, 2); q[3] = c.P(1, 1); q[4] = c.P(1, 2); q[5] = c.P(2, 2);
    q[6] = c.v[0]; q[7] = c.v[1]; q[8] = c.v[2]; q[9] = c.N;
  };
  for (size_t a = 0; a < F; a++) {
    for (int i = 0; i < W; i++) put((*h->voxhess.plvec_voxels[a])[i], clusters + (a * W + i) * 10);
    put(*h->voxhess.sig_vecs[a], fix + a * 10);
    coeffs[a] = h->voxhess.coeffs[a];
  }
  store_poses(h->x_buf, poses);
}

void ref_rw_close(void *hh) { delete (RwHandle *)hh; }

// ---- the octree used INCREMENTALLY: cut_voxel into a live map, one recut per scan, repeated marginalize ------------
// (the calling sequence of consistency.cpp:127-136, repeated; every function called is the reference's)
struct WinHandle {
  std::unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> map;
  int win_count = 0;
  VOX_HESS *vh = nullptr;
  ~WinHandle() { delete vh; for (auto &kv : map) delete kv.second; }
};

void *ref_win_open(int W, double vsize, const float *thr3, int minps, int limit) {
  win_size = W; voxel_size = vsize; min_ps = minps; layer_limit = limit;
  for (int k = 0; k < 3; k++) eigen_value_array[k] = thr3[k];
  return new WinHandle();
}

void ref_win_add_scan(void *hh, const float *xyz, long n, const double *pose12) {
  WinHandle *h = (WinHandle *)hh;
  pcl::PointCloud<PointType> pl;
  pl.reserve((size_t)n);
  for (long k = 0; k < n; k++) { PointType ap; ap.x = xyz[3 * k]; ap.y = xyz[3 * k + 1]; ap.z = xyz[3 * k + 2]; pl.push_back(ap); }
  IMUST x = load_poses(1, pose12)[0];
  cut_voxel(h->map, pl, x, h->win_count);
  h->win_count++;
  for (auto &kv : h->map) kv.second->recut(h->win_count);
}

void ref_win_marginalize(void *hh, int mg, const double *poses) {
  WinHandle *h = (WinHandle *)hh;
  std::vector<IMUST> xs;
  if (poses) xs = load_poses(h->win_count, poses);
  for (auto &kv : h->map) kv.second->marginalize(mg, xs, h->win_count);
  h->win_count -= mg;
}

int ref_win_features(void *hh) {
  WinHandle *h = (WinHandle *)hh;
  delete h->vh;
  h->vh = new VOX_HESS();
  for (auto &kv : h->map) kv.second->tras_opt(*h->vh, h->win_count);
  return (int)h->vh->plvec_voxels.size();
}

// fix clusters: the lower triangle of P (what SelfAdjointEigenSolver reads; P stops being exactly symmetric once
// PointCluster::transform has produced it)
void ref_win_export(void *hh, double *clusters, double *fix, double *coeffs) {
  WinHandle *h = (WinHandle *)hh;
  const int W = win_size;
  const size_t F = h->vh->plvec_voxels.size();
  auto put = [](const PointCluster &c, double *q) {
    q[0] = c.P(0, 0); q[1] = c.P(1, 0); q[2] = c.P(2, 0); q[3] = c.P(1, 1); q[4] = c.P(2, 1); q[5] = c.P(2, 2);
    q[6] = c.v[0]; q[7] = c.v[1]; q[8] = c.v[2]; q[9] = c.N;
  };
  for (size_t a = 0; a < F; a++) {
    for (int i = 0; i < W; i++) put((*h->vh->plvec_voxels[a])[i], clusters + (a * W + i) * 10);
    put(*h->vh->sig_vecs[a], fix + a * 10);
    coeffs[a] = h->vh->coeffs[a];
  }
}

void ref_win_close(void *hh) {
  delete (WinHandle *)hh;
  min_ps = 15; layer_limit = 2;                      // the globals' defaults (bavoxel.hpp:8-12) for whoever runs next
  for (int k = 0; k < 4; k++) eigen_value_array[k] = 1.0 / 16;
}

void ref_exp(const double *w, double *R9) {
  Eigen::Matrix3d R = Exp(Eigen::Vector3d(w[0], w[1], w[2]));
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R9[3 * c + r] = R(r, c);
}
void ref_log(const double *R9, double *w) {
  Eigen::Matrix3d R;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R(r, c) = R9[3 * c + r];
  Eigen::Vector3d l = Log(R);
  w[0] = l[0]; w[1] = l[1]; w[2] = l[2];
}

}  // extern "C"
