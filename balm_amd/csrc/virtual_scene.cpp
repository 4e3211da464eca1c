// ROS/PCL-free restatement of the reference's synthetic-scene driver: the input spec of the hot
// path (host only; no GPU code here).
//   reference: src/benchmark/benchmark_virtual.cpp:547-606 (trajectory, planes, points),
//              :486-503 (initial pose noise), :392-403 (points -> PointCluster per (feature,pose)),
//              :391 (weights = winSize*ptsSize)
// Same distributions, same draw order, std::default_random_engine seeded explicitly (the
// reference seeds with time(0), :547-548).  Points pass through float like pcl::PointXYZINormal
// (:600-602) before they are accumulated in double (:400-401).
//
// Layouts (shared with include/balm_hip.h):
//   pose    12 doubles: R column-major, p
//   cluster 10 doubles: Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N      clusters[(a*W + i)*10 + c]
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace {

struct M3 { double m[3][3]; };
struct V3 { double v[3]; };

M3 ident() { M3 o{}; o.m[0][0] = o.m[1][1] = o.m[2][2] = 1; return o; }

// include/tools.hpp:56-71
M3 rodrigues(const V3 &w) {
  double n = std::sqrt(w.v[0] * w.v[0] + w.v[1] * w.v[1] + w.v[2] * w.v[2]);
  if (n < 1e-11) return ident();
  double x = w.v[0] / n, y = w.v[1] / n, z = w.v[2] / n;
  double K[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
  double s = std::sin(n), c1 = 1.0 - std::cos(n);
  M3 o = ident();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double kk = 0;   // ((1-cos)*K)*K, the reference's evaluation order
      for (int k = 0; k < 3; k++) kk += (c1 * K[r][k]) * K[k][c];
      o.m[r][c] = (o.m[r][c] + s * K[r][c]) + kk;
    }
  return o;
}

V3 mv(const M3 &A, const V3 &x) {
  V3 o;
  for (int r = 0; r < 3; r++) o.v[r] = A.m[r][0] * x.v[0] + A.m[r][1] * x.v[1] + A.m[r][2] * x.v[2];
  return o;
}
V3 mtv(const M3 &A, const V3 &x) {
  V3 o;
  for (int r = 0; r < 3; r++) o.v[r] = A.m[0][r] * x.v[0] + A.m[1][r] * x.v[1] + A.m[2][r] * x.v[2];
  return o;
}
M3 mm(const M3 &A, const M3 &B) {
  M3 o{};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      for (int k = 0; k < 3; k++) o.m[r][c] += A.m[r][k] * B.m[k][c];
  return o;
}

void store_pose(const M3 &R, const V3 &p, double *q) {
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) q[3 * c + r] = R.m[r][c];
  q[9] = p.v[0]; q[10] = p.v[1]; q[11] = p.v[2];
}

struct Dists {
  std::uniform_real_distribution<double> randNorml{-M_PI, M_PI};
  std::uniform_real_distribution<double> randRange;
  std::normal_distribution<double> randThick;
  std::uniform_real_distribution<double> randVoxel{-0.5, 0.5};
  Dists(double surf_range, double point_noise)
      : randRange(-surf_range, surf_range), randThick(0.0, point_noise) {}
};

// one plane feature: benchmark_virtual.cpp:573-606
template <class Engine>
void gen_feature(Engine &e, Dists &d, int a, int W, int pts, const std::vector<M3> &Rs,
                 const std::vector<V3> &ps, double *clusters_a, float *points_a) {
  M3 rot;
  if (a < 3) {
    V3 fd{{0, 0, 0}};
    fd.v[a] = M_PI_2;
    rot = rodrigues(fd);
  } else {
    V3 w;   // braced-init-list evaluation order: left to right
    w.v[0] = d.randNorml(e); w.v[1] = d.randNorml(e); w.v[2] = d.randNorml(e);
    rot = rodrigues(w);
  }
  V3 centre;
  centre.v[0] = d.randRange(e); centre.v[1] = d.randRange(e); centre.v[2] = d.randRange(e);
  for (int j = 0; j < W; j++) {
    double *cl = clusters_a + 10 * (size_t)j;
    for (int c = 0; c < 10; c++) cl[c] = 0;
    for (int k = 0; k < pts; k++) {
      V3 q;
      q.v[0] = d.randVoxel(e); q.v[1] = d.randVoxel(e); q.v[2] = d.randThick(e);
      q = mv(rot, q);
      for (int r = 0; r < 3; r++) q.v[r] += centre.v[r];
      V3 dq{{q.v[0] - ps[j].v[0], q.v[1] - ps[j].v[1], q.v[2] - ps[j].v[2]}};
      V3 b = mtv(Rs[j], dq);
      // pcl::PointXYZINormal stores float32; volatile pins the rounding (g++ 11 -O3 otherwise
      // folds the double->float->double round trip of x and y away inside the SLP-vectorised body)
      volatile float vfx = (float)b.v[0], vfy = (float)b.v[1], vfz = (float)b.v[2];
      const float fx = vfx, fy = vfy, fz = vfz;
      if (points_a) {
        float *o = points_a + 3 * ((size_t)j * pts + k);
        o[0] = fx; o[1] = fy; o[2] = fz;
      }
      double x = fx, y = fy, z = fz;         // tools.hpp:311-316
      cl[0] += x * x; cl[1] += x * y; cl[2] += x * z; cl[3] += y * y; cl[4] += y * z;
      cl[5] += z * z; cl[6] += x; cl[7] += y; cl[8] += z; cl[9] += 1;
    }
  }
}

}  // namespace

extern "C" {

// mode 0: one engine, reference draw order (trajectory -> every feature in order -> pose noise).
// mode 1: trajectory and pose noise from engine(seed); feature a from its own engine(seed+1+a),
//         generated on `threads` host threads (large scenes; not the reference's stream).
// points (optional, may be NULL): F*W*pts*3 floats, body-frame points in (feature, pose, k) order.
// feature_offset (mode 1 only): global index of this call's first feature, so that ranks of a
//         sharded run draw disjoint features against the same trajectory and pose noise.
// coeffs[a] = W*pts (benchmark_virtual.cpp:391).  Returns 0.
int balm_scene_generate(unsigned seed, int W, int F, int pts, double point_noise, double surf_range,
                        int mode, int threads, int feature_offset, double *poses_gt,
                        double *poses_init, double *clusters, double *coeffs, float *points) {
  std::default_random_engine e(seed);
  Dists d(surf_range, point_noise);
  std::normal_distribution<double> rand_traj(-1, 1);   // sic, :559
  V3 rotEnd, traEnd;
  rotEnd.v[0] = rand_traj(e); rotEnd.v[1] = rand_traj(e); rotEnd.v[2] = rand_traj(e);
  traEnd.v[0] = rand_traj(e); traEnd.v[1] = rand_traj(e); traEnd.v[2] = rand_traj(e);
  double nr = std::sqrt(rotEnd.v[0] * rotEnd.v[0] + rotEnd.v[1] * rotEnd.v[1] + rotEnd.v[2] * rotEnd.v[2]);
  double nt = std::sqrt(traEnd.v[0] * traEnd.v[0] + traEnd.v[1] * traEnd.v[1] + traEnd.v[2] * traEnd.v[2]);
  for (int r = 0; r < 3; r++) { rotEnd.v[r] = rotEnd.v[r] / nr * 0.5; traEnd.v[r] = traEnd.v[r] / nt * 1.0; }

  std::vector<M3> Rs(W, ident());
  std::vector<V3> ps(W, V3{{0, 0, 0}});
  for (int i = 1; i < W; i++) {
    double ratio = 1.0 * i / W;
    V3 w{{ratio * rotEnd.v[0], ratio * rotEnd.v[1], ratio * rotEnd.v[2]}};
    Rs[i] = rodrigues(w);
    for (int r = 0; r < 3; r++) ps[i].v[r] = ratio * traEnd.v[r];
  }
  for (int i = 0; i < W; i++) store_pose(Rs[i], ps[i], poses_gt + 12 * (size_t)i);

  const size_t cstride = 10 * (size_t)W, pstride = 3 * (size_t)W * pts;
  if (mode == 0) {
    for (int a = 0; a < F; a++)
      gen_feature(e, d, a, W, pts, Rs, ps, clusters + cstride * a, points ? points + pstride * a : nullptr);
  } else {
    int T = threads < 1 ? 1 : threads;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        Dists dl(surf_range, point_noise);
        for (int a = t; a < F; a += T) {
          const int ga = a + feature_offset;
          std::mt19937_64 ef((uint64_t)seed * 1000003ull + 1 + ga);
          gen_feature(ef, dl, ga, W, pts, Rs, ps, clusters + cstride * a,
                      points ? points + pstride * a : nullptr);
        }
      });
    for (auto &x : th) x.join();
  }
  for (int a = 0; a < F; a++) coeffs[a] = (double)W * pts;

  // benchmark_virtual.cpp:491-503 -- every pose perturbed, pose 0 included
  std::normal_distribution<double> randRot(0, 2 / 57.3);
  std::normal_distribution<double> randTra(0, 0.1);
  for (int i = 0; i < W; i++) {
    V3 rv, tv;
    rv.v[0] = randRot(e); rv.v[1] = randRot(e); rv.v[2] = randRot(e);
    tv.v[0] = randTra(e); tv.v[1] = randTra(e); tv.v[2] = randTra(e);
    for (int r = 0; r < 3; r++) { rv.v[r] /= 1.732; tv.v[r] /= 1.732; }
    M3 Rn = mm(Rs[i], rodrigues(rv));
    V3 pn{{ps[i].v[0] + tv.v[0], ps[i].v[1] + tv.v[1], ps[i].v[2] + tv.v[2]}};
    store_pose(Rn, pn, poses_init + 12 * (size_t)i);
  }
  return 0;
}

// Drop a fraction of observations (sparse co-visibility, like real voxel maps): cluster (a,i) is
// zeroed with probability `drop`, keeping at least `min_obs` observing poses per feature;
// coeffs[a] = sum_i N_i as VOX_HESS::push_voxel does (bavoxel.hpp:42-44).
int balm_scene_sparsify(unsigned seed, int W, int F, double drop, int min_obs, double *clusters,
                        double *coeffs) {
  std::mt19937_64 e(seed);
  std::uniform_real_distribution<double> u(0, 1);
  for (int a = 0; a < F; a++) {
    int kept = W;
    for (int i = 0; i < W; i++) {
      if (u(e) < drop && kept > min_obs) {
        std::memset(clusters + ((size_t)a * W + i) * 10, 0, 80);
        kept--;
      }
    }
    double coe = 0;
    for (int i = 0; i < W; i++) coe += clusters[((size_t)a * W + i) * 10 + 9];
    coeffs[a] = coe;
  }
  return 0;
}

}  // extern "C"
