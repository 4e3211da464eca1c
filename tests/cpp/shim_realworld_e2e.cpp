// The shipped window of datas/benchmark_realworld through the C++ side of the boundary, timed: what a maintainer of the
// reference gets after the one-word patch of INTEGRATION.md 1b -- `BALM2_HIP opt; opt.associate(pl_fulls, x_buf);
// opt.damping_iter(x_buf);` in place of benchmark_realworld.cpp:183-218 -- with the scans held exactly as read_file()
// leaves them (benchmark_realworld.cpp:75-106): one pcl::PointCloud<pcl::PointXYZINormal>::Ptr per scan, 48-byte elements.
//
// Input: a raw window file (written by balm_amd/realworld.py::write_window_bin from datasets/realworld_w177.npz):
//   int32 W, int32 has_ref, int64 n | int64 counts[W] | double poses[W*12] | double ref_poses[W*12] (if has_ref) | float xyz[n*3]
// Output: one line `SHIM_E2E key=value ...` (ms, wall clock, std::chrono::steady_clock around the calls on the caller's thread):
//   cold_create / cold_associate / cold_lm  the first use of the BALM2_HIP object in this process.  Default: the object is declared FIRST
//                                           in main, as INTEGRATION.md 1b recommends -- its constructor starts the device's one-off
//                                           start-up (runtime, pinned ring, code objects) in the background while the scans are read.
//                                           With a fourth argument "late" it is declared after the scans are in memory, right where the
//                                           reference declares `BALM2 opt;` (benchmark_realworld.cpp:217): nothing overlaps.
//   clouds_on_nodes                         how many of the scans' clouds live on NUMA node 0 / 1 / 2 / 3 (where the scheduler ran this reader)
//   associate / lm / total                  median of `reps` further calls on the same object
//   upload / assoc_device                   the library's own HIP-event spans of the median repetition (BALM_T_UPLOAD / BALM_T_VOXEL)
// and, with a third argument, the installed feature table (F, then clusters F*W*10 doubles, coeffs F doubles) as a raw file for
// the bit-exact comparison with the reference's feature set.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sys/syscall.h>
#include <unistd.h>

#include <ros/ros.h>
#include "tools.hpp"
#include "bavoxel.hpp"
#include "balm_shim.hpp"

// the NUMA node a page lives on (move_pages without target nodes only reports); -1: unknown
static int node_of(const void *p) {
  void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095);
  int status = -1;
  return (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) == 0 && status >= 0) ? status : -1;
}

static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: shim_realworld_e2e window.bin [reps] [features_out.bin | -] [late]\n"); return 2; }
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const bool late = argc > 4 && !strcmp(argv[4], "late");
  const char *features_out = (argc > 3 && strcmp(argv[3], "-")) ? argv[3] : nullptr;
  std::unique_ptr<BALM2_HIP> early(late ? nullptr : new BALM2_HIP());       // `BALM2_HIP opt;` as the first statement of main
  FILE *f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  int W = 0, has_ref = 0;
  long long n = 0;
  if (fread(&W, 4, 1, f) != 1 || fread(&has_ref, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1 || W < 1 || n < 1) return 2;
  std::vector<long long> counts((size_t)W);
  std::vector<double> poses((size_t)12 * W), ref((size_t)12 * W);
  if (fread(counts.data(), 8, (size_t)W, f) != (size_t)W || fread(poses.data(), 8, poses.size(), f) != poses.size()) return 2;
  if (has_ref && fread(ref.data(), 8, ref.size(), f) != ref.size()) return 2;
  // read_file(): one cloud per scan, points pushed one by one (benchmark_realworld.cpp:89-96)
  std::vector<pcl::PointCloud<PointType>::Ptr> pl_fulls((size_t)W);
  {
    std::vector<float> buf;
    for (int i = 0; i < W; i++) {
      pl_fulls[(size_t)i].reset(new pcl::PointCloud<PointType>());
      buf.resize((size_t)counts[(size_t)i] * 3);
      if (fread(buf.data(), 4, buf.size(), f) != buf.size()) return 2;
      pl_fulls[(size_t)i]->reserve((size_t)counts[(size_t)i]);
      for (long long k = 0; k < counts[(size_t)i]; k++) {
        PointType ap;
        ap.x = buf[3 * k]; ap.y = buf[3 * k + 1]; ap.z = buf[3 * k + 2];
        ap.intensity = (float)i; ap.curvature = (float)k;
        pl_fulls[(size_t)i]->push_back(ap);
      }
    }
  }
  fclose(f);
  auto to_imust = [&](const double *q) {
    IMUST x;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) x.R(r, c) = q[3 * c + r];
    x.p << q[9], q[10], q[11];
    return x;
  };
  std::vector<IMUST> x_init((size_t)W);
  for (int i = 0; i < W; i++) x_init[(size_t)i] = to_imust(poses.data() + 12 * i);

  // benchmark_realworld.cpp:170,183-185 + launch/benchmark_realworld.launch:4
  win_size = W;
  voxel_size = 2;
  eigen_value_array[0] = 1.0 / 16; eigen_value_array[1] = 1.0 / 16; eigen_value_array[2] = 1.0 / 9;

  struct Row { double associate, lm, upload, assoc_device; };
  std::vector<Row> rows;
  std::vector<IMUST> x_buf;
  int F = 0;
  size_t iters = 0;

  auto t0 = std::chrono::steady_clock::now();
  std::unique_ptr<BALM2_HIP> owner(early ? early.release() : new BALM2_HIP());
  BALM2_HIP &opt = *owner;
  opt.verbose = false;
  opt.timing = true;
  balm_ctx *ctx = opt.context();                       // balm_create: the reference's `BALM2 opt;` costs nothing
  const double cold_create = ms_since(t0);
  double cold_associate = 0, cold_lm = 0;
  for (int rep = 0; rep <= reps; rep++) {
    x_buf = x_init;
    balm_reset_timing(ctx);
    t0 = std::chrono::steady_clock::now();
    F = opt.associate(pl_fulls, x_buf);
    const double ta = ms_since(t0);
    if (F < 3 * W) { printf("SHIM_E2E too few planes: %d\n", F); return 3; }       // benchmark_realworld.cpp:209-215
    t0 = std::chrono::steady_clock::now();
    opt.damping_iter(x_buf);
    const double tl = ms_since(t0);
    double tms[BALM_T_COUNT]; long tcnt[BALM_T_COUNT];
    balm_get_timing(ctx, tms, tcnt);
    iters = opt.last_log.size();
    if (rep == 0) { cold_associate = ta; cold_lm = tl; }
    else rows.push_back({ta, tl, tms[BALM_T_UPLOAD], tms[BALM_T_VOXEL]});
  }
  auto med = [&](double Row::*m) {
    std::vector<double> v;
    for (const Row &r : rows) v.push_back(r.*m);
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  double max_rot = -1, max_tr = -1;
  if (has_ref) {
    max_rot = max_tr = 0;
    for (int i = 0; i < W; i++) {
      IMUST y = to_imust(ref.data() + 12 * i);
      max_rot = std::max(max_rot, Log(y.R.transpose() * x_buf[(size_t)i].R).norm());
      max_tr = std::max(max_tr, (y.p - x_buf[(size_t)i].p).norm());
    }
  }
  if (features_out) {
    std::vector<double> cl((size_t)F * W * 10), co((size_t)F);
    if (balm_get_features(ctx, cl.data(), co.data(), nullptr) != BALM_OK) return 4;
    FILE *o = fopen(features_out, "wb");
    if (!o) return 4;
    long long FF = F;
    fwrite(&FF, 8, 1, o); fwrite(cl.data(), 8, cl.size(), o); fwrite(co.data(), 8, co.size(), o);
    fclose(o);
  }
  size_t npts = 0;
  for (auto &p : pl_fulls) npts += p->size();
  int on_node[4] = {0, 0, 0, 0};
  for (auto &p : pl_fulls) if (!p->empty()) { const int nd = node_of(&p->points[0]); if (nd >= 0 && nd < 4) on_node[nd]++; }
  printf("SHIM_E2E declared=%s clouds_on_nodes=%d/%d/%d/%d scans=%d points=%zu point_bytes=%zu features=%d lm_iterations=%zu reps=%d cold_create=%.3f cold_associate=%.3f cold_lm=%.3f "
         "associate=%.3f lm=%.3f total=%.3f upload=%.3f assoc_device=%.3f max_rot=%.3e max_trans=%.3e\n",
         late ? "late" : "first", on_node[0], on_node[1], on_node[2], on_node[3], W, npts, sizeof(PointType), F, iters, reps, cold_create, cold_associate, cold_lm, med(&Row::associate), med(&Row::lm),
         med(&Row::associate) + med(&Row::lm), med(&Row::upload), med(&Row::assoc_device), max_rot, max_tr);
  return (has_ref && !(max_rot <= 1e-5 && max_tr <= 1e-4)) ? 1 : 0;
}
