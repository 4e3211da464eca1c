// End-to-end drop-in test of include/balm_shim_virtual.hpp (C++ host side of the boundary, virtual flavour).
//
// This translation unit IS the reference's src/benchmark/benchmark_virtual.cpp (included where it lies, its main()
// renamed and not called -- it waits for ROS and a key press) plus the shim header, exactly what a maintainer's
// patched driver would contain.  The driver's own inputs -- one point cloud per plane, pose index in `intensity`
// (:584-599), noisy initial poses (:491-503) -- are drawn by the seeded generator restatement
// (balm_amd/lib/libbalm_scene.so) and go, unchanged, through
//     BALM2      ::dampingIter(x_stats, plSurfs)   the reference's CPU optimizer (:375-482)           and
//     BALM2_HIP  ::dampingIter(x_stats, plSurfs)   include/balm_shim_virtual.hpp -> libbalm_hip.so, MI355X
// and the final poses, the iteration counts and the file's own rsme() are compared (BASELINE.json tolerance:
// 1e-5 rad / 1e-4 m).
//
// Built only where /root/reference exists (tests/cpp/build_shim_driver.sh) into oracle/_ref/; the binary travels to
// the GPU box.  Prints one line "SHIM_VIRTUAL ..." and returns 0 on agreement.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>

#define main balm_reference_benchmark_virtual_main
#include "benchmark_virtual.cpp"
#undef main

#include "balm_shim_virtual.hpp"

typedef int (*gen_fn)(unsigned, int, int, int, double, double, int, int, int, double *, double *, double *, double *,
                      float *);

int main(int argc, char **argv) {
  const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
  const int W = argc > 2 ? atoi(argv[2]) : 20;        // winSize default (:536)
  const int F = argc > 3 ? atoi(argv[3]) : 150;       // sufSize default (:537)
  const int pts = argc > 4 ? atoi(argv[4]) : 40;      // ptsSize default (:538)
  const int ndev = argc > 5 ? atoi(argv[5]) : 1;
  const char *scene_so = argc > 6 ? argv[6] : "balm_amd/lib/libbalm_scene.so";
  void *h = dlopen(scene_so, RTLD_NOW);
  if (!h) { fprintf(stderr, "cannot open %s: %s\n", scene_so, dlerror()); return 2; }
  gen_fn gen = (gen_fn)dlsym(h, "balm_scene_generate");
  std::vector<double> gt(12 * W), init(12 * W), cl((size_t)F * W * 10), co(F);
  std::vector<float> points((size_t)F * W * pts * 3);
  gen(seed, W, F, pts, 0.01, 2.0, 0, 1, 0, gt.data(), init.data(), cl.data(), co.data(), points.data());

  winSize = W; sufSize = F; ptsSize = pts;            // the file's globals (:13-15)
  std::vector<IMUST> xBuf(W);
  xBuf_gt.resize(W);
  for (int i = 0; i < W; i++) {
    const double *q = init.data() + 12 * i, *t = gt.data() + 12 * i;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { xBuf[i].R(r, c) = q[3 * c + r]; xBuf_gt[i].R(r, c) = t[3 * c + r]; }
    xBuf[i].p << q[9], q[10], q[11];
    xBuf_gt[i].p << t[9], t[10], t[11];
  }
  std::vector<pcl::PointCloud<PointType>::Ptr> plSurfs(F);
  PointType ap;
  for (int a = 0; a < F; a++) {
    plSurfs[a].reset(new pcl::PointCloud<PointType>());
    for (int j = 0; j < W; j++) {
      ap.intensity = j;
      for (int k = 0; k < pts; k++) {
        const float *p = points.data() + 3 * (((size_t)a * W + j) * pts + k);
        ap.x = p[0]; ap.y = p[1]; ap.z = p[2];
        plSurfs[a]->push_back(ap);
      }
    }
  }

  // ---- method_test (:505-524): the same clouds and start poses through both classes ----------------------
  std::vector<IMUST> x_ref = xBuf, x_hip = xBuf;
  fflush(stdout);
  BALM2 bm;
  const double t_ref = bm.dampingIter(x_ref, plSurfs);
  BALM2_HIP bh;
  const bool multi = ndev > 1 || getenv("BALM_SHIM_FORCE_MULTI") != nullptr;      // one device through balm_create_multi + RCCL
  bh.n_devices = multi ? ndev : 0;
  const double t_hip = bh.dampingIter(x_hip, plSurfs);
  double rot_ref, tr_ref, rot_hip, tr_hip;
  rsme(x_ref, rot_ref, tr_ref);
  rsme(x_hip, rot_hip, tr_hip);
  double max_rot = 0, max_tr = 0;
  for (int i = 0; i < W; i++) {
    max_rot = std::max(max_rot, Log(x_ref[i].R.transpose() * x_hip[i].R).norm());
    max_tr = std::max(max_tr, (x_ref[i].p - x_hip[i].p).norm());
  }
  const bool anchored = (x_hip[0].R - Eigen::Matrix3d::Identity()).norm() == 0 && x_hip[0].p.norm() == 0;
  printf("SHIM_VIRTUAL W=%d F=%d pts=%d devices=%d iters_hip=%zu max_rot=%.3e max_trans=%.3e rsme_ref=%.6fdeg,%.6fm "
         "rsme_hip=%.6fdeg,%.6fm seconds_ref=%.4f seconds_hip=%.4f anchored=%d multi=%d\n", W, F, pts, ndev, bh.last_log.size(), max_rot,
         max_tr, rot_ref * 57.3, tr_ref, rot_hip * 57.3, tr_hip, t_ref, t_hip, (int)anchored, (int)multi);
  return (max_rot <= 1e-5 && max_tr <= 1e-4 && anchored) ? 0 : 1;
}
