// ORACLE -- TEST INFRASTRUCTURE ONLY.
// Compiles the REFERENCE'S OWN virtual-benchmark translation unit where it lies (never copied):
//     /root/reference/src/benchmark/benchmark_virtual.cpp   (its own copy of class BALM2: u0 = 0.1, 20 iterations,
//                                                            empty fix clusters :377-378, weights winSize*ptsSize :391,
//                                                            cluster build from the point clouds :392-403, warm-up
//                                                            evaluation :405, pose 0 -> identity re-anchor :472-479)
// against the stand-in headers of oracle/compat/ (its main() is renamed and never called; the ROS display code is
// an empty shell), and exposes BALM2::dampingIter on caller-supplied clouds plus the class's evaluators with the flat
// layouts of include/balm_hip.h.  A separate shared object (oracle/_ref/libbalm_ref_virtual.so) because the file
// re-declares class BALM2 of bavoxel.hpp.  This is the copy BASELINE configs[0..3] name.
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>

#define main balm_reference_benchmark_virtual_main
#include "benchmark_virtual.cpp"
#undef main

namespace {

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

// the reference reports progress only by printf (:428); capture stdout to recover the per-iteration rows
struct Capture {
  char path[32]; int fd, saved;
  Capture() { std::strcpy(path, "/tmp/balm_refv_XXXXXX"); fflush(stdout); fd = mkstemp(path); saved = dup(1); dup2(fd, 1); }
  int finish(double *log8, int max_rows) {
    fflush(stdout); dup2(saved, 1); close(saved);
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
};

}  // namespace

extern "C" {

// BALM2::dampingIter(x_stats, plSurfs) (benchmark_virtual.cpp:375-482) on caller-supplied clouds.
// xyz: [F][W][pts][3] float body-frame points, feature-major, pose-major inside a feature (the generator's order,
// :584-599); a point's pose index rides in `intensity` (:586).  Returns the number of log rows; *seconds = the
// reference's own tt2 - tt1.
int refv_damping_iter(int W, int F, int pts, const float *xyz, double *poses, double *log8, int max_rows, double *seconds) {
  ptsSize = pts;        // the file's global: weights are winSize * ptsSize (:391)
  std::vector<pcl::PointCloud<PointType>::Ptr> plSurfs(F);
  PointType ap;
  for (int a = 0; a < F; a++) {
    plSurfs[a].reset(new pcl::PointCloud<PointType>());
    plSurfs[a]->reserve((size_t)W * pts);
    for (int j = 0; j < W; j++) {
      ap.intensity = j;
      for (int k = 0; k < pts; k++) {
        const float *q = xyz + (((size_t)a * W + j) * pts + k) * 3;
        ap.x = q[0]; ap.y = q[1]; ap.z = q[2];
        plSurfs[a]->push_back(ap);
      }
    }
  }
  std::vector<IMUST> xs = load_poses(W, poses);
  Capture cap;
  BALM2 bm;
  const double t = bm.dampingIter(xs, plSurfs);
  const int rows = cap.finish(log8, max_rows);
  store_poses(xs, poses);
  if (seconds) *seconds = t;
  return rows;
}

// the class's own evaluators on flat clusters: form 0 left_evaluate_acc2 (:219-339), form 1 accEvaluate2 (:110-217)
int refv_evaluate(int form, int W, int F, const double *clusters, const double *fix, const double *coeffs,
                  const double *poses, double *Hess, double *JacT, double *residual) {
  BALM2 bm;
  bm.winSize = W;
  for (int a = 0; a < F; a++) {
    auto *v = new std::vector<PointCluster>(W);
    for (int i = 0; i < W; i++) {
      const double *q = clusters + ((size_t)a * W + i) * 10;
      PointCluster &c = (*v)[i];
      c.P << q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5];
      c.v << q[6], q[7], q[8];
      c.N = (int)q[9];
    }
    PointCluster *fx = new PointCluster();
    if (fix) {
      const double *q = fix + (size_t)a * 10;
      fx->P << q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5];
      fx->v << q[6], q[7], q[8];
      fx->N = (int)q[9];
    }
    bm.plvecVoxels.push_back(v);
    bm.sig_vecs.push_back(fx);
    bm.coeffs.push_back(coeffs[a]);
  }
  std::vector<IMUST> xs = load_poses(W, poses);
  Eigen::MatrixXd H(6 * W, 6 * W);
  Eigen::VectorXd J(6 * W);
  double r = 0;
  int rc = 0;
  if (form == 0) bm.left_evaluate_acc2(xs, H, J, r);
  else if (form == 1) bm.accEvaluate2(xs, H, J, r);
  else if (form == 3) bm.only_residual(xs, r);
  else rc = 1;
  if (Hess && form != 3) std::memcpy(Hess, H.data(), sizeof(double) * 36 * W * W);
  if (JacT && form != 3) std::memcpy(JacT, J.data(), sizeof(double) * 6 * W);
  *residual = r;
  for (auto p : bm.plvecVoxels) delete p;
  for (auto p : bm.sig_vecs) delete p;
  return rc;
}

// rsme() (:48-62) against the file's global ground truth
void refv_rsme(int W, const double *gt, const double *es, double *rot, double *tran) {
  xBuf_gt = load_poses(W, gt);
  std::vector<IMUST> xe = load_poses(W, es);
  rsme(xe, *rot, *tran);
}

}  // extern "C"
