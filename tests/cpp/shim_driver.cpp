// End-to-end drop-in test of include/balm_shim.hpp (C++ host side of the boundary).
//
// Mirrors the body of the reference's benchmark_realworld main (src/benchmark/benchmark_realworld.cpp:
// 163-218) with synthetic plane clouds in place of the PCD files: the reference's OWN, unmodified
// association code (cut_voxel -> OCTO_TREE_ROOT::recut -> tras_opt -> VOX_HESS::push_voxel, compiled
// from /root/reference/src/benchmark/bavoxel.hpp) builds the VOX_HESS feature container, then the
// same container and the same initial poses go through
//     BALM2      ::damping_iter   (the reference's CPU optimizer, bavoxel.hpp:1069)  and
//     BALM2_HIP  ::damping_iter   (include/balm_shim.hpp -> libbalm_hip.so, MI355X)
// and the final poses are compared (BASELINE.json tolerance: 1e-5 rad / 1e-4 m).
//
// Built only where /root/reference exists (tests/cpp/build_shim_driver.sh) into oracle/_ref/; the
// binary travels to the GPU box.  Prints one line: "SHIM_DRIVER features=.. iters_ref=.. iters_hip=..
// max_rot=.. max_trans=.. resid_rel=..".
#include <dlfcn.h>
#include <ros/ros.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tools.hpp"
#include "bavoxel.hpp"
#include "balm_shim.hpp"

typedef int (*gen_fn)(unsigned, int, int, int, double, double, int, int, int, double *, double *, double *, double *,
                      float *);

int main(int argc, char **argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
  const int W = argc > 2 ? atoi(argv[2]) : 20;
  const int F = argc > 3 ? atoi(argv[3]) : 150;     // sufSize default of the reference (benchmark_virtual.cpp:538)
  const int pts = argc > 4 ? atoi(argv[4]) : 40;
  const char *scene_so = argc > 5 ? argv[5] : "balm_amd/lib/libbalm_scene.so";
  void *h = dlopen(scene_so, RTLD_NOW);
  if (!h) { fprintf(stderr, "cannot open %s: %s\n", scene_so, dlerror()); return 2; }
  gen_fn gen = (gen_fn)dlsym(h, "balm_scene_generate");
  std::vector<double> gt(12 * W), init(12 * W), cl((size_t)F * W * 10), co(F);
  std::vector<float> points((size_t)F * W * pts * 3);
  // planes spread over +-15 m so that 1 m voxels rarely hold two of them (the generator's default
  // +-2 m packs 150 patches into 64 voxels)
  gen(seed, W, F, pts, 0.01, 15.0, 0, 1, 0, gt.data(), init.data(), cl.data(), co.data(), points.data());

  // poses and per-frame clouds, exactly what read_file() would hand over (benchmark_realworld.cpp:75-106)
  // initial poses: a fifth of the generator's noise (0.4 deg / 2 cm), an odometry-grade start like the
  // shipped alidarPose.csv, so that the adaptive voxelisation still sees planes at 15 m range
  std::vector<IMUST> x_buf(W);
  for (int i = 0; i < W; i++) {
    const double *q = init.data() + 12 * i, *t = gt.data() + 12 * i;
    Eigen::Matrix3d Ri, Rg;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { Ri(r, c) = q[3 * c + r]; Rg(r, c) = t[3 * c + r]; }
    Eigen::Vector3d pi(q[9], q[10], q[11]), pg(t[9], t[10], t[11]);
    x_buf[i].R = Rg * Exp(0.2 * Log(Rg.transpose() * Ri));
    x_buf[i].p = pg + 0.2 * (pi - pg);
  }
  std::vector<pcl::PointCloud<PointType>::Ptr> pl_fulls(W);
  for (int i = 0; i < W; i++) pl_fulls[i].reset(new pcl::PointCloud<PointType>());
  for (int a = 0; a < F; a++)
    for (int i = 0; i < W; i++)
      for (int k = 0; k < pts; k++) {
        const float *p = points.data() + 3 * (((size_t)a * W + i) * pts + k);
        PointType ap; ap.x = p[0]; ap.y = p[1]; ap.z = p[2]; ap.intensity = i;
        pl_fulls[i]->push_back(ap);
      }

  // ---- benchmark_realworld.cpp:163-200, verbatim structure ------------------------------------
  IMUST es0 = x_buf[0];
  for (uint i = 0; i < x_buf.size(); i++) {
    x_buf[i].p = es0.R.transpose() * (x_buf[i].p - es0.p);
    x_buf[i].R = es0.R.transpose() * x_buf[i].R;
  }
  win_size = x_buf.size();
  voxel_size = 1;
  unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> surf_map;
  eigen_value_array[0] = 1.0 / 16;
  eigen_value_array[1] = 1.0 / 16;
  eigen_value_array[2] = 1.0 / 9;
  for (int i = 0; i < win_size; i++) cut_voxel(surf_map, *pl_fulls[i], x_buf[i], i);
  VOX_HESS voxhess;
  for (auto iter = surf_map.begin(); iter != surf_map.end(); iter++) {
    iter->second->recut(win_size);
    iter->second->tras_opt(voxhess, win_size);
  }
  const size_t nfeat = voxhess.plvec_voxels.size();
  if (nfeat < 3 * x_buf.size()) { printf("SHIM_DRIVER too few planes: %zu\n", nfeat); return 3; }

  // ---- the same container through both optimizers ----------------------------------------------
  std::vector<IMUST> x_ref = x_buf, x_hip = x_buf, x_ab(W);
  fflush(stdout);
  BALM2 opt_ref;
  opt_ref.damping_iter(x_ref, voxhess);                       // reference, CPU
  BALM2_HIP opt_hip;
  opt_hip.damping_iter(x_hip, voxhess);                       // shim -> libbalm_hip.so, GPU

  // ---- association on the device too: scans -> BALM2_HIP::associate -> damping_iter -----------------
  std::vector<IMUST> x_dev = x_buf;
  BALM2_HIP opt_dev;
  opt_dev.verbose = false;
  const int nfeat_dev = opt_dev.associate(pl_fulls, x_dev);
  opt_dev.damping_iter(x_dev);
  double dev_rot = 0, dev_tr = 0;
  for (int i = 0; i < W; i++) {
    dev_rot = std::max(dev_rot, Log(x_ref[i].R.transpose() * x_dev[i].R).norm());
    dev_tr = std::max(dev_tr, (x_ref[i].p - x_dev[i].p).norm());
  }

  // ---- the map used incrementally: cut_voxel + recut per scan, tras_opt, marginalize(2, poses), two more scans ----
  // (the reference's own octree against BALM2_HIP::window_*; the "new" scans are scans 0 and 1 seen again)
  unordered_map<VOXEL_LOC, OCTO_TREE_ROOT *> live_map;
  BALM2_HIP opt_win;
  opt_win.verbose = false;
  opt_win.window_open();
  int wc = 0;
  auto push_scan = [&](int i) {
    cut_voxel(live_map, *pl_fulls[i], x_buf[i], wc);
    wc++;
    for (auto &kv : live_map) kv.second->recut(wc);
    opt_win.window_add_scan(*pl_fulls[i], x_buf[i]);
  };
  for (int i = 0; i < W; i++) push_scan(i);
  std::vector<IMUST> x_opt = x_ref;                      // "optimised" poses: the reference's result
  for (auto &kv : live_map) kv.second->marginalize(2, x_opt, wc);
  wc -= 2;
  opt_win.window_marginalize(2, x_opt);
  push_scan(0); push_scan(1);
  VOX_HESS vh_win;
  for (auto &kv : live_map) kv.second->tras_opt(vh_win, wc);
  const int nfeat_win = opt_win.window_features();
  double coe_ref = 0, fix_ref = 0, coe_dev = 0, fix_dev = 0;
  for (size_t a = 0; a < vh_win.coeffs.size(); a++) { coe_ref += vh_win.coeffs[a]; fix_ref += vh_win.sig_vecs[a]->N; }
  {
    std::vector<double> co((size_t)nfeat_win), fx((size_t)nfeat_win * 10);
    balm_get_features(opt_win.context(), nullptr, co.data(), nullptr);
    balm_get_association(opt_win.context(), fx.data(), nullptr);
    for (int a = 0; a < nfeat_win; a++) { coe_dev += co[(size_t)a]; fix_dev += fx[(size_t)a * 10 + 9]; }
  }
  const bool win_ok = (size_t)nfeat_win == vh_win.coeffs.size() && coe_ref == coe_dev && fix_ref == fix_dev && fix_ref > 0;
  for (auto &kv : live_map) delete kv.second;

  double max_rot = 0, max_tr = 0;
  for (int i = 0; i < W; i++) {
    Eigen::Vector3d l = Log(x_ref[i].R.transpose() * x_hip[i].R);
    max_rot = std::max(max_rot, l.norm());
    max_tr = std::max(max_tr, (x_ref[i].p - x_hip[i].p).norm());
  }
  // and the evaluators through the reference's divide_thread_left signature
  Eigen::MatrixXd H1(6 * W, 6 * W), H2;
  Eigen::VectorXd J1(6 * W), J2;
  double r1 = opt_ref.divide_thread_left(x_buf, voxhess, x_ab, H1, J1);
  double r2 = opt_hip.divide_thread_left(x_buf, voxhess, x_ab, H2, J2);
  double hmax = 0, hdiff = 0;
  for (int c = 0; c < 6 * W; c++) for (int r = 0; r < 6 * W; r++) {
    hmax = std::max(hmax, std::fabs(H1(r, c))); hdiff = std::max(hdiff, std::fabs(H1(r, c) - H2(r, c)));
  }
  printf("SHIM_DRIVER features=%zu iters_hip=%zu max_rot=%.3e max_trans=%.3e resid_rel=%.3e hess_rel=%.3e "
         "dev_features=%d dev_rot=%.3e dev_trans=%.3e win_features=%d/%zu win_points=%.0f/%.0f win_fix_points=%.0f/%.0f\n", nfeat,
         opt_hip.last_log.size(), max_rot, max_tr, std::fabs(r1 - r2) / r1, hdiff / hmax, nfeat_dev, dev_rot, dev_tr, nfeat_win,
         vh_win.coeffs.size(), coe_dev, coe_ref, fix_dev, fix_ref);
  for (auto &kv : surf_map) delete kv.second;
  return (max_rot <= 1e-5 && max_tr <= 1e-4 && hdiff / hmax < 1e-10 && (size_t)nfeat_dev == nfeat && dev_rot <= 1e-5 &&
          dev_tr <= 1e-4 && win_ok) ? 0 : 1;
}
