// balm_shim.hpp -- host-side mirror of the reference's optimizer interface on top of the C ABI
// (include/balm_hip.h).  Header-only C++14; needs what the reference's own translation unit
// already has in scope when it uses BALM2: tools.hpp (IMUST, PointCluster, DVEL), Eigen
// (MatrixXd / VectorXd), the global `int win_size` and class VOX_HESS (the feature container that
// OCTO_TREE_NODE::tras_opt fills through VOX_HESS::push_voxel, src/benchmark/bavoxel.hpp:30-51,920).
//
// It provides class BALM2_HIP with the public interface of class BALM2
// (src/benchmark/bavoxel.hpp:984-1168): same method names, argument order and meaning, in-place pose
// update, progress printf line and "too few planes" behaviour -- the C++ voxel-association code and
// the benchmark drivers call it unchanged (one-word change at the declaration: `BALM2_HIP opt;`
// instead of `BALM2 opt;`).  All arithmetic runs in libbalm_hip.so on the GPU;
// there is no CPU fallback: a missing library / GPU aborts with a message.
#ifndef BALM_SHIM_HPP
#define BALM_SHIM_HPP

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "balm_hip.h"

class BALM2_HIP {
 public:
  // knobs the reference hard-codes inside damping_iter (bavoxel.hpp:1087,1104,1155)
  double u0 = 0.01;
  int max_iter = 10;
  double rel_tol = 1e-6;
  int min_planes_per_pose = 20;
  int form = BALM_FORM_LEFT;   // bavoxel.hpp:1109 (left) vs :1108 (right, commented out there)
  int device = 0;
  int n_devices = 0;           // >= 1: balm_create_multi over devices device..device+n_devices-1 (features sharded, one RCCL
                               // all-reduce per evaluation inside the library; replaces the thread sum at bavoxel.hpp:1049-1056)
  bool timing = false;         // BALM_FLAG_TIMING: HIP-event spans per kernel class (balm_get_timing on context())
  bool int8_syrk = false;      // BALM_FLAG_SYRK_INT8 (opt-in): the dense Hessian products on the INT8 matrix cores, FP64-exact to ~1e-12
  bool verbose = true;         // the reference always prints its per-iteration line (:1132)
  bool reanchor = true;        // bavoxel.hpp:1159-1164 (the consistency driver does not: BAs_left.hpp:1087)
  double abs_tol = 0;          // > 0: the consistency driver's stop rule |r1-r2| < 1e-9 (BAs_left.hpp:1083)
  std::vector<balm_iter_log> last_log;

  // The device's one-off start-up (runtime, pinned upload ring, code objects) begins in the background here: declare the object
  // before the scans are read and the first associate() / damping_iter() finds a warm device.  `dev`: as the `device` member.
  explicit BALM2_HIP(int dev = 0) : device(dev) { balm_prewarm(dev); }
  ~BALM2_HIP() { if (ctx_) balm_destroy(ctx_); }
  BALM2_HIP(const BALM2_HIP &) = delete;
  BALM2_HIP &operator=(const BALM2_HIP &) = delete;

  // bavoxel.hpp:989
  double divide_thread_right(std::vector<IMUST> &x_stats, VOX_HESS &voxhess, std::vector<IMUST> &x_ab,
                             Eigen::MatrixXd &Hess, Eigen::VectorXd &JacT) {
    (void)x_ab;
    return evaluate(BALM_FORM_RIGHT, x_stats, voxhess, Hess, JacT);
  }

  // bavoxel.hpp:1025
  double divide_thread_left(std::vector<IMUST> &x_stats, VOX_HESS &voxhess, std::vector<IMUST> &x_ab,
                            Eigen::MatrixXd &Hess, Eigen::VectorXd &JacT) {
    (void)x_ab;
    return evaluate(BALM_FORM_LEFT, x_stats, voxhess, Hess, JacT);
  }

  // bavoxel.hpp:1061
  double only_residual(std::vector<IMUST> &x_stats, VOX_HESS &voxhess, std::vector<IMUST> &x_ab) {
    (void)x_ab;
    upload(voxhess);
    std::vector<double> poses = flatten_poses(x_stats);
    double r = 0;
    check(balm_only_residual(ctx_, poses.data(), &r));
    return r;
  }

  // Replaces the association block of the reference's drivers (benchmark_realworld.cpp:183-200: cut_voxel per
  // scan into the surf_map, OCTO_TREE_ROOT::recut, ::tras_opt -> VOX_HESS::push_voxel) with one device call.
  // Reads the same globals the reference's code reads: win_size, voxel_size, eigen_value_array, min_ps
  // (bavoxel.hpp:8-17; layer_limit 0..2).  `CloudPtr` is pcl::PointCloud<PointType>::Ptr (any pointer to a
  // container whose `points` is a contiguous vector of elements that start with float x, y, z).  The clouds are read
  // where they lie by the library's host threads.  Returns the number of plane features, which are installed on the
  // device: follow with damping_iter(x_stats).
  template <class CloudPtr>
  int associate(const std::vector<CloudPtr> &pl_fulls, const std::vector<IMUST> &x_buf) {
    ensure_ctx();
    if (layer_limit < 0 || layer_limit > 2 || (int)pl_fulls.size() != win_size || (int)x_buf.size() != win_size) {
      fprintf(stderr, "balm_hip: associate needs layer_limit in 0..2 and win_size scans and poses\n");
      abort();
    }
    // the clouds stay where they are: the library's host threads pack x, y, z out of the 48-byte PointType elements straight into
    // its pinned upload chunks (balm_associate_scans); the caller's thread only lists W pointers and counts
    std::vector<const void *> base((size_t)win_size);
    std::vector<long> count((size_t)win_size);
    for (int i = 0; i < win_size; i++) {
      const auto &pts = pl_fulls[(size_t)i]->points;
      static_assert(offsetof(typename std::decay<decltype(pts[0])>::type, x) == 0, "PointType: x, y, z lead the element");
      base[(size_t)i] = pts.empty() ? nullptr : (const void *)&pts[0];
      count[(size_t)i] = (long)pts.size();
    }
    std::vector<double> poses = flatten_poses(x_buf);
    balm_voxel_opts o;
    balm_voxel_defaults(&o);
    o.voxel_size = voxel_size;
    for (int l = 0; l < 3; l++) o.eigen_thr[l] = eigen_value_array[l];
    o.min_ps = min_ps;
    o.layer_limit = layer_limit;
    int F = 0;
    long roots = 0;
    check(balm_associate_scans(ctx_, &o, win_size, base.data(), count.data(), sizeof(pl_fulls[0]->points[0]), poses.data(), &F, &roots));
    loaded_ = (const void *)this;        // the device features no longer mirror a VOX_HESS
    loaded_F_ = (size_t)F;
    return F;
  }

  // The map used incrementally, on the device (balm_window_*): the three loops a sliding-window caller writes over its
  // unordered_map<VOXEL_LOC, OCTO_TREE_ROOT*> --
  //     cut_voxel(surf_map, pl, x, win_count - 1);  for (roots) root->recut(win_count);        -> window_add_scan(pl, x)
  //     for (roots) root->tras_opt(voxhess, win_count);  opt.damping_iter(x_buf, voxhess);      -> window_features(); damping_iter(x_buf)
  //     for (roots) root->marginalize(mg, x_buf, win_count);                                    -> window_marginalize(mg, x_buf)
  // with the globals voxel_size, eigen_value_array, min_ps, layer_limit, win_size read as the reference reads them.
  // Set reanchor = false for windows that carry fix clusters (they are world-frame; src/simulation/BAs_left.hpp's
  // damping_iter does not re-anchor either).
  void window_open() {
    ensure_ctx();
    balm_voxel_opts o;
    balm_voxel_defaults(&o);
    o.voxel_size = voxel_size;
    for (int l = 0; l < 3; l++) o.eigen_thr[l] = eigen_value_array[l];
    o.min_ps = min_ps;
    o.layer_limit = layer_limit;
    check(balm_window_open(ctx_, &o));
  }
  template <class Cloud>
  void window_add_scan(const Cloud &pl, const IMUST &x) {
    std::vector<double> pose = flatten_poses(std::vector<IMUST>(1, x));
    if (pl.points.empty()) { fprintf(stderr, "balm_hip: window_add_scan: empty scan\n"); abort(); }
    check(balm_window_add_scan_strided(ctx_, &pl.points[0], (long)pl.points.size(), sizeof(pl.points[0]), pose.data()));
  }
  int window_features() {
    int F = 0;
    check(balm_window_features(ctx_, &F));
    loaded_ = (const void *)this;
    loaded_F_ = (size_t)F;
    return F;
  }
  void window_marginalize(int mg_size, const std::vector<IMUST> &x_poses) {      // an empty x_poses = no re-transform
    std::vector<double> poses = flatten_poses(x_poses);
    check(balm_window_marginalize(ctx_, mg_size, x_poses.empty() ? nullptr : poses.data()));
  }

#ifdef POINT_NOISE
  // The consistency driver's optimizer (src/simulation/BAs_left.hpp:1025-1098; compiled when the translation unit
  // includes src/simulation/toolss.hpp, whose PointCluster carries the 9x9 noise covariance c_cov): LM loop with
  // that file's constants (u0 = 0.01, up to 1000 iterations, stop at |r1 - r2| < 1e-9, no re-anchoring, no plane
  // precheck), then `Rcov = Hess^-1 (sum Ls c_cov Ls^T) Hess^-T` at the result (:1089-1096).
  void damping_iter(std::vector<IMUST> &x_stats, VOX_HESS &voxhess, Eigen::MatrixXd &Rcov, int covEnable = 1) {
    upload(voxhess, /*force=*/true);
    const int keep_iter = max_iter, keep_planes = min_planes_per_pose;
    const double keep_rel = rel_tol, keep_abs = abs_tol;
    const bool keep_anchor = reanchor;
    max_iter = 1000; min_planes_per_pose = 0; rel_tol = 0; abs_tol = 1e-9; reanchor = false;
    run_lm(x_stats);
    max_iter = keep_iter; min_planes_per_pose = keep_planes; rel_tol = keep_rel; abs_tol = keep_abs; reanchor = keep_anchor;
    if (!covEnable) return;
    printf("Begin to compute covariance matrix...\n");          // :1091
    const size_t F = voxhess.plvec_voxels.size();
    const int W = win_size, n = 6 * W;
    std::vector<double> cc(F * (size_t)W * 81);
    for (size_t a = 0; a < F; a++)
      for (int i = 0; i < W; i++) {
        const PointCluster &c = (*voxhess.plvec_voxels[a])[(size_t)i];
        double *q = cc.data() + (a * W + i) * 81;
        for (int r = 0; r < 9; r++)
          for (int k = 0; k < 9; k++) q[9 * r + k] = c.c_cov(r, k);
      }
    std::vector<double> poses = flatten_poses(x_stats);
    Rcov.resize(n, n);
    check(balm_pose_covariance(ctx_, poses.data(), cc.data(), 0.0, Rcov.data(), nullptr));
  }
#endif

  balm_ctx *context() { ensure_ctx(); return ctx_; }     // for the C ABI's read-back calls (balm_get_features, ...)

  // damping_iter on the features installed by associate() / window_features()
  void damping_iter(std::vector<IMUST> &x_stats) { run_lm(x_stats); }

  // bavoxel.hpp:1069
  void damping_iter(std::vector<IMUST> &x_stats, VOX_HESS &voxhess) {
    upload(voxhess, /*force=*/true);      // one upload per optimisation: the container may have been refilled
    run_lm(x_stats);
  }

 private:
  void run_lm(std::vector<IMUST> &x_stats) {
    std::vector<double> poses = flatten_poses(x_stats);
    balm_lm_opts o;
    o.form = form; o.u0 = u0; o.max_iter = max_iter; o.rel_tol = rel_tol;
    o.min_planes_per_pose = min_planes_per_pose; o.force_hess = 0; o.no_stop = 0;
    o.verbose = verbose ? 1 : 0; o.reanchor = reanchor ? 1 : 0; o.abs_tol = abs_tol;
    last_log.assign((size_t)max_iter, balm_iter_log());
    int iters = 0;
    int rc = balm_damping_iter(ctx_, &o, poses.data(), last_log.data(), &iters);
    if (rc == BALM_ERR_TOO_FEW_PLANES) {      // bavoxel.hpp:1079-1085, verbatim behaviour
      printf("Initial error too large.\n");
      printf("Please loose plane determination criteria for more planes.\n");
      printf("The optimization is terminated.\n");
      exit(0);
    }
    check(rc);
    last_log.resize((size_t)iters);
    for (size_t i = 0; i < x_stats.size(); i++) {
      const double *q = poses.data() + 12 * i;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) x_stats[i].R(r, c) = q[3 * c + r];
      x_stats[i].p << q[9], q[10], q[11];
    }
  }

  balm_ctx *ctx_ = nullptr;
  int ctx_win_ = 0;
  const void *loaded_ = nullptr;
  size_t loaded_F_ = 0;

  void check(int rc) {
    if (rc == BALM_OK) return;
    fprintf(stderr, "balm_hip: %s (code %d)\n", ctx_ ? balm_last_error(ctx_) : "no context", rc);
    abort();
  }

  void ensure_ctx() {
    if (ctx_ && ctx_win_ == win_size) return;
    if (ctx_) balm_destroy(ctx_);
    const int flags = (timing ? BALM_FLAG_TIMING : 0) | (int8_syrk ? BALM_FLAG_SYRK_INT8 : 0);
    ctx_ = n_devices >= 1 ? balm_create_multi(win_size, device, n_devices, flags) : balm_create(win_size, device, flags);
    ctx_win_ = win_size;
    loaded_ = nullptr;
    if (!ctx_) {
      fprintf(stderr, "balm_hip: balm_create(win_size=%d, device=%d) failed: no MI355X / libbalm_hip.so?\n",
              win_size, device);
      abort();
    }
  }

  // VOX_HESS holds borrowed pointers (bavoxel.hpp:24-26); flatten them into the ABI's arrays.  The
  // evaluators (divide_thread_*, only_residual) reuse the upload while they are called on the same
  // container object of the same size, as the reference's damping_iter does within one optimisation.
  void upload(VOX_HESS &vh, bool force = false) {
    ensure_ctx();
    const size_t F = vh.plvec_voxels.size();
    if (!force && loaded_ == (const void *)&vh && loaded_F_ == F) return;
    const int W = win_size;
    std::vector<double> fx(F * 10), co(F);
    bool any_fix = false;
    for (size_t a = 0; a < F; a++) {
      put(*vh.sig_vecs[a], fx.data() + a * 10);
      any_fix = any_fix || vh.sig_vecs[a]->N != 0;
      co[a] = vh.coeffs[a];
    }
    // the F x W clusters go from the octree's own vectors straight into the library's pinned upload chunks, pulled by its
    // host threads (balm_set_features_cb): no flattened 80-bytes-per-cluster copy of the table on the way
    struct Src { const VOX_HESS *vh; int W; } src{&vh, W};
    check(balm_set_features_cb(ctx_, (int)F, [](void *user, int f0, int f1, double *dst) {
      const Src &s = *static_cast<const Src *>(user);
      for (int a = f0; a < f1; a++) {
        const std::vector<PointCluster> &v = *s.vh->plvec_voxels[(size_t)a];
        for (int i = 0; i < s.W; i++) put(v[(size_t)i], dst + ((size_t)(a - f0) * s.W + i) * 10);
      }
    }, &src, any_fix ? fx.data() : nullptr, co.data()));
    loaded_ = (const void *)&vh;
    loaded_F_ = F;
  }

  static void put(const PointCluster &c, double *q) {
    q[0] = c.P(0, 0); q[1] = c.P(0, 1); q[2] = c.P(0, 2); q[3] = c.P(1, 1); q[4] = c.P(1, 2); q[5] = c.P(2, 2);
    q[6] = c.v[0]; q[7] = c.v[1]; q[8] = c.v[2]; q[9] = (double)c.N;
  }

  static std::vector<double> flatten_poses(const std::vector<IMUST> &xs) {
    std::vector<double> p(12 * xs.size());
    for (size_t i = 0; i < xs.size(); i++) {
      double *q = p.data() + 12 * i;
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) q[3 * c + r] = xs[i].R(r, c);
      q[9] = xs[i].p[0]; q[10] = xs[i].p[1]; q[11] = xs[i].p[2];
    }
    return p;
  }

  double evaluate(int f, std::vector<IMUST> &x_stats, VOX_HESS &voxhess, Eigen::MatrixXd &Hess, Eigen::VectorXd &JacT) {
    upload(voxhess);
    std::vector<double> poses = flatten_poses(x_stats);
    const int n = 6 * win_size;
    Hess.resize(n, n);
    JacT.resize(n);
    double r = 0;
    check(balm_evaluate(ctx_, f, poses.data(), 0, (int)voxhess.plvec_voxels.size(), Hess.data(), JacT.data(), &r));
    return r;
  }
};

#endif  // BALM_SHIM_HPP
