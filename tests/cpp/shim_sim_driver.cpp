// Drop-in test of include/balm_shim.hpp against the CONSISTENCY driver's interface (N4).
//
// Mirrors the optimizer call of src/simulation/consistency.cpp:150-156: a VOX_HESS of noisy clusters (with their
// 9x9 noise covariances, accumulated by the reference's own PointCluster::push, toolss.hpp:315-347) and fix
// clusters goes through
//     BALM2     ::damping_iter(x_buf, voxhess, Rcov)   (the reference, src/simulation/BAs_left.hpp:1025)   and
//     BALM2_HIP ::damping_iter(x_buf, voxhess, Rcov)   (include/balm_shim.hpp -> libbalm_hip.so, MI355X);
// poses, the covariance and the NEES of consistency.cpp:159-170 are compared.
// Built only where /root/reference exists (tests/cpp/build_shim_driver.sh) into oracle/_ref/shim_sim_driver.
#include <dlfcn.h>
#include <ros/ros.h>

#include <cstdio>
#include <cstdlib>
#include <random>

#include "toolss.hpp"
#include "BAs_left.hpp"
#include "balm_shim.hpp"

typedef int (*gen_fn)(unsigned, int, int, int, double, double, int, int, int, double *, double *, double *, double *,
                      float *);

int main(int argc, char **argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
  const int W = argc > 2 ? atoi(argv[2]) : 8;
  const int F = argc > 3 ? atoi(argv[3]) : 40;
  const int pts = argc > 4 ? atoi(argv[4]) : 30;
  const char *scene_so = argc > 5 ? argv[5] : "balm_amd/lib/libbalm_scene.so";
  void *h = dlopen(scene_so, RTLD_NOW);
  if (!h) { fprintf(stderr, "cannot open %s: %s\n", scene_so, dlerror()); return 2; }
  gen_fn gen = (gen_fn)dlsym(h, "balm_scene_generate");
  const int WA = W + 1;                                    // pose 0 is the marginalised scan -> fix clusters
  std::vector<double> gt(12 * WA), init(12 * WA), cl((size_t)F * WA * 10), co(F);
  std::vector<float> points((size_t)F * WA * pts * 3);
  gen(seed, WA, F, pts, 0.0, 2.0, 0, 1, 0, gt.data(), init.data(), cl.data(), co.data(), points.data());   // noise-free points

  win_size = W;
  pnoise = 0.02;
  std::default_random_engine e(seed);
  std::normal_distribution<double> noise(0.0, pnoise);
  std::vector<IMUST> x_gt(WA);
  for (int i = 0; i < WA; i++) {
    const double *q = gt.data() + 12 * i;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) x_gt[i].R(r, c) = q[3 * c + r];
    x_gt[i].p << q[9], q[10], q[11];
  }
  VOX_HESS voxhess;
  std::vector<std::vector<PointCluster> *> owned;
  std::vector<PointCluster *> fixes;
  for (int a = 0; a < F; a++) {
    auto *v = new std::vector<PointCluster>(W);
    PointCluster *fx = new PointCluster();
    for (int i = 0; i < WA; i++)
      for (int k = 0; k < pts; k++) {
        const float *p = points.data() + 3 * (((size_t)a * WA + i) * pts + k);
        Eigen::Vector3d pv(p[0], p[1], p[2]);
        if (i == 0) { fx->push(x_gt[0].R * pv + x_gt[0].p); continue; }
        if ((a + i) % 5 == 0) continue;                    // ragged co-visibility
        pv[0] += noise(e); pv[1] += noise(e); pv[2] += noise(e);          // OCTO_TREE_NODE::corrupt (BAs_left.hpp:886-906)
        (*v)[i - 1].push(pv);
      }
    owned.push_back(v); fixes.push_back(fx);
    voxhess.push_voxel(v, fx, 0, 0);
  }
  std::vector<IMUST> x_true(x_gt.begin() + 1, x_gt.end()), x_ref = x_true, x_hip = x_true;
  Eigen::MatrixXd Rcov_ref(6 * W, 6 * W), Rcov_hip;
  Rcov_ref.setZero();
  fflush(stdout);
  BALM2 opt_ref;
  opt_ref.damping_iter(x_ref, voxhess, Rcov_ref);
  BALM2_HIP opt_hip;
  opt_hip.damping_iter(x_hip, voxhess, Rcov_hip);

  double max_rot = 0, max_tr = 0, cmax = 0, cdiff = 0;
  for (int i = 0; i < W; i++) {
    max_rot = std::max(max_rot, Log(x_ref[i].R.transpose() * x_hip[i].R).norm());
    max_tr = std::max(max_tr, (x_ref[i].p - x_hip[i].p).norm());
  }
  for (int c = 0; c < 6 * W; c++) for (int r = 0; r < 6 * W; r++) {
    cmax = std::max(cmax, std::fabs(Rcov_ref(r, c))); cdiff = std::max(cdiff, std::fabs(Rcov_ref(r, c) - Rcov_hip(r, c)));
  }
  // consistency.cpp:159-170 with either result
  double nees[2];
  for (int which = 0; which < 2; which++) {
    const std::vector<IMUST> &xe = which ? x_hip : x_ref;
    Eigen::VectorXd err(6 * W); err.setZero();
    for (int i = 0; i < W; i++) {
      err.block<3, 1>(6 * i, 0) = Log(x_true[i].R * xe[i].R.transpose());
      err.block<3, 1>(6 * i + 3, 0) = -x_true[i].R * xe[i].R.transpose() * xe[i].p + x_true[i].p;
    }
    const Eigen::MatrixXd &Rc = which ? Rcov_hip : Rcov_ref;
    Eigen::VectorXd sol = Rc.inverse() * err;
    nees[which] = err.dot(sol);
  }
  printf("SHIM_SIM_DRIVER W=%d features=%d iters_hip=%zu max_rot=%.3e max_trans=%.3e cov_rel=%.3e nees_ref=%.2f nees_hip=%.2f expected=%d\n",
         W, F, opt_hip.last_log.size(), max_rot, max_tr, cdiff / cmax, nees[0], nees[1], 6 * W);
  for (auto p : owned) delete p;
  for (auto p : fixes) delete p;
  return (max_rot <= 1e-5 && max_tr <= 1e-4 && cdiff / cmax < 1e-6 && std::fabs(nees[0] - nees[1]) < 1e-4 * nees[0]) ? 0 : 1;
}
